/* msclip_hip.h -- C ABI of libmsclip_hip.so (gfx950 / MI355X).
 *
 * The reference (Hxyou/MSCLIP) is pure Python: it has no FFI for this path, the
 * boundary is an nn.Module.  These entry points are what the
 * product's Python modules bind with ctypes; each one names the reference code
 * it replaces ("M.py" = lib/models/clip_openai_pe_res_v1.py).
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller; kernels
 * never allocate, never synchronise, and are ordered on `stream` (a
 * hipStream_t passed as void*; NULL = the legacy default stream).  bf16 is the
 * raw upper half of an IEEE fp32 (uint16).  Return value: 0 = launched,
 * MSCLIP_EINVAL (-1) = rejected arguments, MSCLIP_ELAUNCH (-2) = HIP launch
 * error.  "out_kind": 0 = bf16, 1 = fp32.
 */
#ifndef MSCLIP_HIP_H
#define MSCLIP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* out[row(m), n] = epi(alpha * sum_k X[m,k] * W[n,k])  -- bf16 MFMA GEMM.
 * Replaces every F.linear / 1x1 / 3x3 conv on the path: M.py:612 (QKV in-proj),
 * :747 (out_proj), :794-798 (MLP), :1993-2000 + :1920-1936 (stem stages, BN and the
 * 1x1 shortcut folded into the 3x3 weight), :1842-1861 (bottleneck convs),
 * :1573-1586 (adapter 1x1), :2690/:3074 (projections), :3141 (logits).
 *   mode 0: X dense row-major [M, K], leading dimension ldx (elements, %8 == 0).
 *   mode 1: X NHWC bf16 [B, H, Wd, Cin]; M = B*Ho*Wo; K-chunk c (8 channels) of a row is
 *           described by ktab[c] = doff | kh<<20 | kw<<24 (negative = zero padding chunk),
 *           doff = (kh*Wd + kw)*Cin + ch; taps outside the image read `zero`.
 *   W is [N][ldw] bf16 with K % 64 == 0 (zero padded), ldw % 8 == 0.  N, ldo, ldr multiples of 4 take
 *   the vectorised epilogue, anything else an element-wise tail path.
 *   epilogue: v = alpha*acc + bias[n]; act 1 = QuickGELU (M.py:222-224); then
 *   + resid (1: fp32 [m][n], 2: bf16 [m][n], 3: fp32 table row (m % rpg + roff), e.g.
 *   positional embedding); act 2 = ReLU (after the residual); store to row
 *   m + (m / rpg) * radd + roff (token scatter of M.py:2418-2425).
 *   Training-step forms (ping-pong kernel only, dense X, N % 8 == 0): out2 != NULL additionally stores v BEFORE the
 *   activation (bf16, leading dimension ldo) -- c_fc writes the pre-activation the backward needs and QuickGELU(v) in one
 *   launch; resid_kind 4: v *= QuickGELU'(resid) with resid bf16 [m][n] -- the dgrad GEMM of c_proj applies the activation's
 *   derivative (d/dh h sigma(1.702 h)) on the saved pre-activation in its epilogue.
 *   resid_kind 5 (mode 1, generic / streaming kernels): v = resid[store row][n] > 0 ? v : 0 with resid bf16 -- ReLU backward on the
 *   saved activation, which is laid out like the output (same row scatter, ldr); the conv side's input gradients.
 *   resid_kind 6 (bf16 out, generic / streaming kernels): v += resid[store row][n] (bf16) -- a scattered launch accumulating into
 *   an existing map (resid == out: in place); the stride-2 shortcuts' input gradients into their block's main-path gradient. */
typedef struct msclip_gemm_desc {
  const void* X;
  const void* W;
  const void* zero;      /* >= 16 bytes of zeros */
  void* out;
  const float* bias;     /* [N] or NULL */
  const void* resid;     /* or NULL */
  int M, N, K;
  int ldx, ldw, ldo, ldr;  /* leading dimensions in elements; ldw >= K */
  int mode;
  int H, Wd, Cin, Ho, Wo, stride, pad;
  const int* ktab;
  int act;
  int resid_kind;
  int out_kind;
  float alpha;
  int rpg, radd, roff;   /* rpg > 0; use rpg = INT_MAX, radd = roff = 0 for the identity */
  void* out2;            /* optional second bf16 output: the epilogue value before the activation (NULL: none) */
  float out_scale;       /* out_kind 2 (e4m3 output, msclip_gemm_f8 only): stored value = fp8(epilogue value * out_scale), saturating */
  int tile;              /* 0 = auto, 1 = 128x128 (4 waves), 4 = 256x256 ping-pong (what auto picks for large problems; dense X, or implicit conv with Cin % 64 == 0), 5 = streaming kernel with LDS-resident weights (K <= 192, or the 3x3 convolutions with 48 input channels; M >= 4096; auto picks it there), 6 = 256x192 two-buffer (implicit conv with N % 192 == 0; on dense operands: the main loop a fused in_proj + attention kernel would run, kept for its calibration -- tools/probes/fused_calib.py), (7 and 8, the two epilogue-hiding kernels of rounds 2-3, were measured slower and retired in round 4: EINVAL) */
  int wg_cap;            /* 0 = the launch may take every CU; > 0: at most this many persistent workgroups (CUs) for the dense / implicit-conv MFMA kernels (round-4 probe of two half-batch chains on two streams, tools/probes/two_chain_probe.py: measured neutral, the engine leaves it 0) */
  /* ---- LayerNorm fold (dense ping-pong kernel only; M, seg_split and N multiples of 256; DESIGN.md "LayerNorm fold").
   * Consumer (a projection that follows a LayerNorm, M.py:1027-1028 + 204-219): X holds bf16 (x - center[m]) instead of the
   * LayerNorm output, W carries gamma (W'[n][k] = gamma[k] W[n][k]), and
   *     out[m][n] = act( rowstat[m][0] * acc - rowstat[m][1] * csum[n] + bias[n] ),
   * rowstat[m] = (rstd, (mean - center) * rstd) of row m, csum[n] = sum_k W'[n][k], bias = b + W beta.  Rows >= seg_split (the
   * other modality's tokens: own gamma / beta, shared W) take W2 / bias2 / csum2.  bf16 output, act 0 / 1, no residual. */
  const void* W2;        /* NULL: one row segment */
  const float* bias2;
  const float* csum;     /* [N] (with rowstat) */
  const float* csum2;
  const float* rowstat;  /* [M][2] fp32; NULL: fold off */
  int seg_split;
  /* Producer (out_proj / c_proj, resid_kind 1, fp32 out): besides out = resid + acc + bias it writes
   * xb[m][n] = bf16(out[m][n] - center[m]) and part[m][n / 64] = (sum, sum of squares) of (out - center) over 64 columns. */
  int ldxb;
  void* xb;              /* bf16 [M][ldxb]; NULL: off */
  const float* center;   /* [M] */
  float* part;           /* [M][N / 64][2].  Without xb, resid_kind 4 only (the training step's dgrad launch that writes
                          * dh = (dY W) * QuickGELU'(h)): fp32 [M / 128][N], row r = the column sums of the values stored to
                          * out over rows [128 r, 128 r + 128) -- the caller folds them (msclip_colsum) into c_fc's bias
                          * gradient instead of re-reading out; NULL: off */
  const void* resid2;    /* producer only: rows >= seg_split read their residual from resid2[m][ldr] (m the absolute row) instead of
                          * resid -- the image rows' stream sits in the lateral adapter's output buffer behind an adapter (M.py:1777)
                          * while the text rows' is `out` itself; NULL: every row from resid */
  const int* M_dev;      /* dense ping-pong kernel (msclip_gemm's "pp" variant, msclip_gemm_f8) only: when not NULL the kernel
                          * runs min(M, *M_dev) rows -- the row count of a packed caption batch lives on the device
                          * (msclip_text_lengths' dims), M is the upper bound the launch is sized and validated for; the value
                          * must satisfy the same divisibility rules as M (whole 256-row tiles for the fold forms) */
  /* Train-mode BatchNorm in two passes over the convolution's INPUT instead of a raw fp32 output map (round 6; streaming kernel
   * only = msclip_gemm_variant "stream": pointwise / 1x1 stride-2 convolutions and the 3x3 ones over 48 input channels,
   * M.py:1812-1861, 1898-1936 in train()).  bn_mode 1 (statistics): nothing is stored; every wave of the launch leaves the column
   * sums of its rows, part[wave][0][n] = sum x, part[wave][1][n] = sum x^2 (fp32 [part_rows][2][N], part_rows >= the launch's
   * waves is enforced by shrinking the launch; rows of waves that do not exist stay untouched: zero them, fold with msclip_colsum).
   * bn_mode 2 (normalise): bn_consts [5][N] = msclip_bn_finish's output rows (mean, variance, rstd, scale, shift): out = act(x scale +
   * shift [+ resid, resid_kind 2]) (bf16) and out2 = xhat = (x - mean) rstd (bf16 [M][ldo]) -- what the backward reads instead of
   * the raw map.  bias must be NULL, alpha 1, N % 8 == 0, no row scatter. */
  int bn_mode;
  int part_rows;
  const float* bn_consts;
} msclip_gemm_desc;

int msclip_gemm(const msclip_gemm_desc* desc, void* stream);

/* Split-K form of the dense GEMM for deep, narrow contractions (weight gradients): out is fp32 [slices][M][ldo], slice s
 * holds the contraction over K columns [s*K/slices, (s+1)*K/slices) (K % (64*slices) == 0); no bias / residual /
 * activation / scatter.  The caller folds the slices (msclip_colsum over [slices, M*ldo]: fixed order). */
int msclip_gemm_splitk(const msclip_gemm_desc* desc, int slices, void* stream);

/* The same contraction for TOKEN-major operands -- what a weight gradient dW = dY^T X is made of -- without
 * transposing them first: desc->X is [T, ldx] bf16 whose COLUMNS are the output rows m (dY), desc->W is [T, ldw] whose columns
 * are the output columns n (the layer input), desc->K = T tokens (any count: rows past T contribute zero), out fp32
 * [slices][M][ldo] partials (slice s = token rows [s * ceil(T / 64 / slices) * 64, ...)).  The ping-pong kernel with 64-token x
 * 128-channel LDS regions and ds_read_b64_tr_b16 fragment reads.  No bias / residual / activation. */
int msclip_gemm_splitk_tn(const msclip_gemm_desc* desc, int slices, void* stream);

/* The dense ping-pong GEMM on OCP e4m3 (fp8) operands, CDNA4's MX matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 with
 * unit block scales (BASELINE config C5; the reference has no fp8 semantics).  desc->X [M, K] and desc->W [N, K] are e4m3 BYTES
 * (ldx / ldw in bytes, multiples of 16; K a multiple of 128; mode 0 only); row_scale [M] and col_scale [N] are the fp32
 * per-row scales of X and W: out = epilogue(alpha * row_scale[m] * col_scale[n] * sum_k X[m, k] W[n, k]), same bias /
 * activation / residual / output kinds as msclip_gemm, plus out_kind 2: an e4m3 output scaled by desc->out_scale (no residual;
 * M % 256 == 0, N and ldo multiples of 16) -- c_fc writes the MLP hidden matrix as the fp8 operand of c_proj with one static,
 * calibrated scale per tensor.  Replaces F.linear (M.py:612, 794, 798) when MODEL.SPEC.PRECISION is fp8. */
int msclip_gemm_f8(const msclip_gemm_desc* desc, const float* row_scale, const float* col_scale, void* stream);

/* LayerNorm (M.py:204-219; parameters (gamma, beta) for rows < split, (gamma2, beta2) from there on) straight to e4m3 with one
 * fp32 scale per row: q[m][c] = fp8(y / s[m]), s[m] = max_c |y| / 448.  The quantising producer of msclip_gemm_f8's X operand. */
int msclip_layernorm_f8(const float* x, int ldx, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                        int split, void* q, int ldq, float* row_scale, int M, int C, float eps, const int* m_dev, void* stream);

/* bf16 [M, C] rows -> e4m3 + per-row scale (same convention).  C % 8 == 0. */
int msclip_quant_f8_rows(const void* x, int ldx, void* q, int ldq, float* row_scale, int M, int C, void* stream);

/* Name of the kernel msclip_gemm would launch for this descriptor ("pp", "ppconv", "stream", "dense128",
 * "conv192", "conv128"; "invalid" for rejected arguments): the library's own dispatch rule, so
 * that measurement code (bench.py's roofline leg) counts exactly the launches of one kernel.  No GPU work. */
const char* msclip_gemm_variant(const msclip_gemm_desc* desc);

/* Fused softmax(q k^T [+ causal]) v per (sample, head), head_dim 64, L <= 288 (257 = the 16 x 16 grid of a 14-pixel patch); q pre-scaled.
 * qkv: bf16 [nsamples*L, ldq] with columns [q | k | v], each heads*64 wide.  out: bf16
 * [nsamples*L, ldo], ldq % 8 == 0 and ldo % 8 == 0 (16-byte row pieces).  Replaces M.py:707-738 (scale, reshapes, bmm, mask add, softmax, bmm). */
int msclip_attention(const void* qkv, void* out, int nsamples, int L, int heads, int ldq, int ldo, int causal,
                     void* stream);

/* out[m] = LayerNorm(x[src(m)]) with fp32 statistics, eps inside the sqrt (M.py:204-219).
 * src(m) = row_idx ? row_idx[m] : m*row_mul + row_add.  C in {256, 512, 768}.  raw_out (optional)
 * receives the un-normalised fp32 row as well (moves a residual row between buffers for free). */
int msclip_layernorm(const float* x, int ldx, const int* row_idx, int row_mul, int row_add, const float* gamma,
                     const float* beta, void* out, int ldo, int out_kind, float* raw_out, int ld_raw, int M, int C,
                     float eps, void* stream);

/* msclip_layernorm over contiguous rows that also leaves what the LayerNorm fold needs about these rows: center[m] = the row's
 * mean (the next producing GEMM centres its bf16 copy on it) and rowstat[m] = (1, 0) (the consuming GEMM takes `out` as it
 * is).  Either pointer may be NULL.  m_dev (optional, here and in msclip_rowstat_finalize / msclip_layernorm_f8): a device int;
 * the kernel runs min(M, *m_dev) rows (packed captions: the row count never visits the host, M sizes the launch). */
int msclip_layernorm_stats(const float* x, int ldx, const float* gamma, const float* beta, void* out, int ldo, int out_kind,
                           float* raw_out, int ld_raw, float* center, float* rowstat, int M, int C, float eps, const int* m_dev,
                           void* stream);

/* Row statistics of a producing GEMM's partial sums (msclip_gemm_desc.part): for rows [0, M)
 *   mu = sum_g part[m][g][0] / C,  var = sum_g part[m][g][1] / C - mu^2  (sums of x - center[m]: the subtraction is benign),
 *   rowstat[m] = (1 / sqrt(var + eps), mu / sqrt(var + eps)),  center[m] += mu.   groups = C / 64, folded in index order. */
int msclip_rowstat_finalize(const float* part, int groups, float* center, float* rowstat, int M, int C, float eps, const int* m_dev,
                            void* stream);

/* The same over one [M, C] matrix whose rows [0, split) and [split, M) carry different parameters: the image and
 * the text tokens of the shared residual matrix with their modality-specific LayerNorms (M.py:1027-1028 run per
 * tower) in one launch. */
int msclip_layernorm_split(const float* x, int ldx, const float* gamma, const float* beta, const float* gamma2,
                           const float* beta2, int split, void* out, int ldo, int out_kind, int M, int C, float eps,
                           void* stream);

/* x[row_base + b*L + l] = emb[tokens[b,l]] + pos[l]; eot_row[b] = row_base + b*L + argmax_l tokens[b,l]
 * (M.py:3047-3048 and the EOT pick of :3057-3060).  tokens are int64. */
int msclip_embed_tokens(const long long* tokens, const float* emb, const float* pos, float* x, int ldx, int* eot_row,
                        int B, int L, int C, int vocab, int row_base, void* stream);

/* ---- Packed (pad-free) captions.  The causal mask (M.py:2965-2971) lets a token see only earlier positions, and encode_text
 * returns the row at the EOT position argmax_l tokens[b,l] (M.py:3057-3060): rows behind that position cannot influence any
 * output (or receive any gradient) in ANY block.  Caption b therefore owns n_b = argmax_l tokens[b,l] + 1 consecutive rows of
 * the token matrix, at cu[b] = sum_{j<b} n_j, instead of L.
 * msclip_text_lengths: len[b] = n_b (first maximum, like torch.argmax); cu[0..B-1] as above, cu[B] = total live rows,
 * cu[B+1] = max_b n_b (cu holds B + 2 ints); eot_row[b] = row_base + cu[b] + n_b - 1 (optional).  tokens int64 [B, L].
 * dims (optional, 8 ints): the row counts the launches over the text rows read ON THE DEVICE (msclip_gemm_desc.M_dev, the m_dev /
 * rows_dev / dims arguments below), so that neither the host nor a hipGraph capture ever needs the batch's total:
 *   dims[0] = total live rows, [1] = longest caption, [2] = padded = total rounded up to a multiple of pad_to (pad_to 0: total),
 *   [3] = row_base + padded (rows of the whole token matrix when the text segment starts at row_base), [4] = padded - total,
 *   [5] = row_base + total, [6] = row_base.  cap_rows >= B * L = the rows the caller's buffers hold behind row_base. */
int msclip_text_lengths(const long long* tokens, int B, int L, int row_base, int* len, int* cu, int* eot_row, int* dims, int pad_to,
                        int cap_rows, void* stream);

/* x[row_base + cu[b] + l] = emb[tokens[b,l]] + pos[l] for l < n_b (M.py:3047-3048 on the live rows); the rows
 * [row_base + cu[B], row_base + rows_padded) -- tile padding of the GEMMs over the packed segment, < 256 rows -- are zeroed.
 * rows_dev (optional): device int, the padded row count (dims + 2 of msclip_text_lengths); rows_padded is then its upper bound. */
int msclip_embed_tokens_packed(const long long* tokens, const float* emb, const float* pos, float* x, int ldx, const int* cu,
                               int B, int L, int C, int vocab, int row_base, int rows_padded, const int* rows_dev, void* stream);

/* msclip_attention over packed captions: sample b's tokens are rows cu[b] .. cu[b+1] of qkv / out (both start at the text
 * segment), Lmax >= every length (<= 96: picks the tile count).  The pad_rows (< 256) output rows behind cu[nsamples] are
 * zeroed.  Replaces M.py:707-738 with the mask of :2965-2971 on the rows that can matter.  dims (optional): msclip_text_lengths'
 * device block; the padding rows zeroed are then min(pad_rows, dims[4]) (pad_rows = the launch's bound, 255). */
int msclip_attention_varlen(const void* qkv, void* out, const int* cu, int nsamples, int Lmax, int heads, int ldq, int ldo,
                            int causal, int pad_rows, const int* dims, void* stream);

/* ---- Fused in_proj + attention (BASELINE.json north_star: "fused QKV-projection + SDPA"; replaces F.linear with in_proj_weight /
 * in_proj_bias, M.py:612, and the attention core of M.py:707-738 with the causal mask of :2965-2971 for captions) in ONE kernel:
 * q|k|v never reach HBM.  Opt-in (engine: MSCLIP_FUSED_QKV_ATTN=1): measured slower than the ping-pong GEMM + attention launches
 * (DESIGN.md s0 item 4).  A tile = whole samples, at most 256 token rows, x one head's 192 columns [q | k | v]:
 *   W      bf16 [heads * 192, ldw]: the packed in_proj weight re-ordered head-major, rows h * 192 + {0..63: q_h (pre-scaled by
 *          head_dim^-0.5), 64..127: k_h, 128..191: v_h}; bias (and csum) fp32 [heads * 192] in the same order, 16-byte aligned;
 *   cu     [nsamples + 1] first row of every sample in X / out (ascending; cu[nsamples] = end of the live rows);
 *   rowseg [M][2] (first row, end row) of the sample that owns each live row; tile_first [ntiles + 1] first sample of every tile
 *          (both from msclip_qkvattn_tables with max_rows = 256); ntiles_dev (optional) = device copy of the tile count, read by the kernel when the
 *          host does not know it (packed captions);
 *   rows of samples that start at or behind causal_from_row attend causally (INT_MAX: none; 0: all);
 *   LayerNorm fold (optional, as msclip_gemm's consumer form): rowstat [M][2], csum; rows >= seg_split take W2 / bias2 / csum2.
 * out bf16 [M, ldo] receives the attention output (heads concatenated) of the live rows; samples of up to 96 rows. */
typedef struct msclip_qkvattn_desc {
  const void* X;
  const void* W;
  const void* zero;      /* >= 16 bytes of zeros */
  void* out;
  const float* bias;
  const int* cu;
  const int* tile_first;
  const int* rowseg;
  const int* ntiles_dev; /* or NULL: ntiles below */
  int M, K, heads, ntiles;
  int ldx, ldw, ldo;
  int causal_from_row;
  const float* rowstat;  /* NULL: X holds the LayerNorm output itself */
  const float* csum;
  const void* W2;        /* NULL: one row segment */
  const float* bias2;
  const float* csum2;
  int seg_split;
} msclip_qkvattn_desc;

int msclip_qkv_attention(const msclip_qkvattn_desc* desc, void* stream);

/* rowseg / tile_first / *ntiles of msclip_qkv_attention from the samples' first rows cu [nsamples + 1]: whole samples packed
 * greedily into tiles of at most max_rows (256 or 128) rows, no tile straddling sample index split_sample (the image / text boundary; <= 0 or
 * >= nsamples: none).  *ntiles = -1 when more than max_tiles tiles would be needed (tile_first holds max_tiles + 1 ints). */
int msclip_qkvattn_tables(const int* cu, int nsamples, int split_sample, int* rowseg, int* tile_first, int* ntiles, int max_tiles,
                          int max_rows, void* stream);

/* msclip_attention_lastq over packed captions: sample b's keys are the rows row_base + cu[b] .. row_base + cu[b+1] of qkv
 * (all of them: the query is the caption's last live row, the EOT position). */
int msclip_attention_lastq_varlen(const void* q, int ldqc, const void* qkv, int ldq, void* out, int ldo, int nsamples, int Lmax,
                                  int heads, const int* cu, int row_base, void* stream);

/* msclip_attention_bwd over packed captions (Lmax <= 96); the pad_rows rows of dqkv behind cu[nsamples] are zeroed (the
 * in_proj weight gradient contracts over them).  colsum_part as in msclip_attention_bwd. */
int msclip_attention_bwd_varlen(const void* qkv, const void* o, const void* dout, void* dqkv, const int* cu, int nsamples, int Lmax,
                                int heads, int ldq, int ldo, int causal, int pad_rows, float* colsum_part, void* stream);

/* msclip_embed_tokens_bwd over packed captions: dx row cu[b] + l belongs to tokens[b,l].  dEmb[token] += row (fp32 atomics);
 * dPos[l] = sum over the captions with n_b > l of their row l, in caption order (bitwise repeatable; dpos is overwritten,
 * rows l >= max n_b become zero; NULL: not formed). */
int msclip_embed_tokens_bwd_packed(const long long* tokens, const float* dx, int lddx, const int* cu, float* demb, float* dpos,
                                   int B, int L, int C, int vocab, void* stream);

/* x[b*L] = class_embedding + positional_embedding[0] (M.py:2421-2425). */
int msclip_fill_cls(const float* cls, const float* pos, float* x, int ldx, int B, int L, int C, void* stream);

/* Lateral_Adapter.forward bottom half + sum + ln_adapt (M.py:1763-1777):
 * xout = LN([cls; BN(dw3x3(grid))] + [cls; t]); dww is [9][C] with BN folded, t fp32 [B*g*g, ldt]. */
int msclip_adapter_combine_ln(const float* xin, int ldx, const float* t, int ldt, const float* dww, const float* dwb,
                              const float* gamma, const float* beta, float* xout, int ldo, int B, int L, int g, int C,
                              int usecls, float eps, void* stream);

/* The same pass followed by the transformer block's ln_1 of each row while it is still in registers (M.py:1777 then :1027):
 * lno = bf16 LN(xout[m]; gamma1, beta1), and the LayerNorm fold's per-row state as msclip_layernorm_stats leaves it
 * (center[m] = mean of xout[m], rowstat[m] = (1, 0)).  xout stays the block's fp32 residual stream for these rows. */
int msclip_adapter_combine_ln_stats(const float* xin, int ldx, const float* t, int ldt, const float* dww, const float* dwb,
                                    const float* gamma, const float* beta, float* xout, int ldo, const float* gamma1,
                                    const float* beta1, void* lno, int ldl, float* center, float* rowstat, int B, int L, int g,
                                    int C, int usecls, float eps, void* stream);

/* y = x / ||x||_2 (M.py:2983, :3076); writes fp32 and/or bf16 copies. */
int msclip_l2norm(const float* x, int ldx, float* out_f32, int ldf, void* out_bf16, int ldb, int M, int E,
                  void* stream);

/* out[m] = x[src(m)], rows as raw bytes (row_bytes, both leading dimensions and both pointers multiples of 16).
 * src(m) = row_idx ? row_idx[m] : m*row_mul + row_add.  Moves the rows that are still read after the last block's
 * attention -- x[:, 0, :] of the image tower (M.py:2685) and the EOT row of every caption (M.py:3057-3060) -- into a compact
 * matrix, so that the block's out_proj / ln_2 / MLP run on 2 B rows instead of all tokens. */
int msclip_gather_rows(const void* x, long long ldx_bytes, const int* row_idx, int row_mul, int row_add, void* out,
                       long long ldo_bytes, int M, int row_bytes, void* stream);

/* Both Cin=3 3x3/s2/p1 convs (stem conv1+bn1+relu, M.py:1993-1995; parallel stage 0, M.py:2260-2273)
 * in one pass over the NCHW image.  w: fp32 [27][2*C1] (BN folded), bias [2*C1]; outputs NHWC bf16. */
int msclip_stem_conv3x3s2_dual(const void* img, int img_is_bf16, const float* w, const float* bias, void* out_a,
                               void* out_b, int B, int H, int W, int C1, void* stream);

/* The same pass WITHOUT BatchNorm fold, bias and ReLU, fp32 outputs [B * Ho * Wo][48] each: the raw convolution outputs that
 * train-mode BatchNorm normalises with batch statistics (the training step's forward; nn.BatchNorm2d in train(), M.py:1993-1995,
 * 2260-2273).  w: fp32 [27][96], row ci * 9 + kh * 3 + kw, columns = conv1's 48 output channels then parallel stage 0's; image and
 * filters are rounded to bf16 for the MFMA like every convolution operand here.  No patch matrix. */
int msclip_stem_conv3x3s2_dual_raw(const void* img, int img_is_bf16, const float* w, float* out_a, float* out_b, int B, int H, int W,
                                   void* stream);
/* Round 6: the two-pass form of train-mode BatchNorm over the same two convolutions (M.py:1939-1946 conv1 + bn1; :2260-2273 the
 * parallel branch's stage 0) -- the raw maps are never written.
 * msclip_stem_conv3x3s2_dual_stats: part [part_waves][2][2][48] fp32 = per wave (sum x, sum x^2) of conv a, then conv b (part_waves
 *   a multiple of 4 = 4 x the launch's workgroups; fold the rows with msclip_colsum).
 * msclip_stem_conv3x3s2_dual_norm: the convolutions again; stats_a / stats_b fp32 [5][48] = msclip_bn_finish's output rows (mean,
 *   variance, rstd, scale, shift) per convolution: y = relu(x scale + shift) (the expression of msclip_bn_apply), xhat = (x - mean)
 *   rstd, all four outputs bf16 [B * H/2 * W/2][48].  The backward reads xhat where it read the raw fp32 map (mean 0, rstd 1,
 *   gamma := scale). */
int msclip_stem_conv3x3s2_dual_stats(const void* img, int img_is_bf16, const float* w, float* part, int part_waves, int B, int H, int W,
                                     void* stream);
int msclip_stem_conv3x3s2_dual_norm(const void* img, int img_is_bf16, const float* w, const float* stats_a, const float* stats_b,
                                    void* y_a, void* y_b, void* xhat_a, void* xhat_b, int B, int H, int W, void* stream);

/* The same pass fused with the 3x3/s2/p1 convolution that consumes branch a (stem resnet_stage.conv_0 with its folded
 * 1x1 shortcut and ReLU, M.py:1920-1936): branch a's 48-channel map stays in LDS (8x8 output tiles, 17x17 windows),
 * branch b (parallel stage 0) is written as before.  w2: bf16 [Cout][448] (K = (kh*3+kw)*48 + c, zero padded),
 * b2 [Cout], out2 NHWC bf16 [B, Ho/2.., Cout]; C1 is 48, Cout 48 or 96. */
int msclip_stem_dual_conv3x3s2(const void* img, int img_is_bf16, const float* w, const float* bias, void* out_b,
                               const void* w2, const float* b2, void* out2, int B, int H, int W, int Cout,
                               void* stream);

/* Patch matrix of a kernel == stride == P convolution over an NCHW image [B, 3, H, W] (fp32 or bf16), the plain patch conv
 * of M.py:2502-2508 / 2657 as a dense GEMM operand: out[b*g*g + py*g + px][c*P*P + kh*P + kw] (bf16, row stride kpad, columns
 * [3*P*P, kpad) zero; kpad % 64 == 0).  The convolution itself is msclip_gemm over this matrix with the token scatter /
 * positional-table epilogue (rpg = g*g, radd = roff = 1). */
int msclip_patchify(const void* img, int img_is_bf16, void* out, int kpad, int B, int H, int W, int P, void* stream);

/* relu(conv3x3/s2/p1(relu(conv1x1(x)))) with folded BatchNorms (ConvResBlock conv1-bn1-relu-conv2-bn2-relu,
 * M.py:1825-1840) without materialising the 1x1's output.  x NHWC bf16 [B, H, W, 48]; w1 bf16 [48][64] (K padded),
 * w2 bf16 [Cout][448]; out NHWC bf16 [B, (H-1)/2+1, (W-1)/2+1, Cout]; Cout 48 or 96. */
int msclip_conv1x1_conv3x3s2(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                             void* out, int B, int H, int W, int Cout, void* stream);

/* A whole ConvResBlock with projection shortcut, 48 -> (48 mid) -> 96 channels at stride 2 (the first stage of the
 * parallel branch, M.py:1825-1861): relu(conv3(relu(conv2(relu(conv1 x)))) + shortcut(x)), BatchNorms folded.  conv1's
 * map, conv2's output and the shortcut's output stay on chip; conv3, the strided 1x1 shortcut and the residual add run
 * as one K = 96 contraction per output pixel.  w1 bf16 [48][64], w2 bf16 [48][448], w3 / wr bf16 [96][64] (K = 48
 * zero padded), b3r = conv3 bias + shortcut bias [96]; x NHWC bf16 [B, H, W, 48]; out NHWC bf16 [B, Ho, Wo, 96]. */
int msclip_convresblock48_s2(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                             const void* w3, const void* wr, const float* b3r, void* out, int B, int H, int W,
                             void* stream);

/* Depthwise kernel==stride conv of the adapters (M.py:1573-1581): NHWC bf16 in, [B*g*g, ldo] bf16 out,
 * w fp32 [k*k][C] with the BN scale folded (the BN shift goes into the following 1x1's bias). */
int msclip_dwpool(const void* top, const float* w, void* out, int ldo, int B, int H, int W, int C, int k,
                  void* stream);

/* Fused contrastive reduction: per-row (max, sum-exp) partials of scale * A[R,E] @ B[N,E]^T over `nsplit` column
 * ranges, and the label logit diag[i] = scale * A[i] . B[label_off + i]; the [R, N] logits are never written
 * (MFMA GEMM + online log-sum-exp; the logits themselves are M.py:3141, the loss is not in the reference).
 * A, B: bf16, E % 16 == 0, E <= 512.  part_max / part_sum: fp32 [R, nsplit]. */
int msclip_clip_lse_fused(const void* A, int lda, const void* B, int ldb, int R, int N, int E, float scale,
                          int label_off, int nsplit, float* part_max, float* part_sum, float* diag, void* stream);

/* out[0] = scale * sum_i (LSE_img[i] - diag[i]) + (LSE_txt[i] - diag[i]) from the partials of the image->text and
 * text->image sweeps; lse_out (optional, fp32 [2, R]) receives the merged LSEs. */
int msclip_clip_loss_from_partials(const float* pmax_img, const float* psum_img, const float* pmax_txt,
                                   const float* psum_txt, const float* diag, int R, int nsplit, float scale, float* out,
                                   float* lse_out, void* stream);

/* lse[r] = log sum_n exp(logits[r, n]) over a fp32 block [R, N]. */
int msclip_lse_rows(const float* logits, int ld, float* lse, int R, int N, void* stream);

/* out[0] = scale * sum_i (lse_img[i] - d_i) + (lse_txt[i] - d_i), d_i = img_rows[i, label_off + i].
 * With scale = 1/(2*N_global) the sum over ranks is the symmetric CLIP cross-entropy. */
int msclip_clip_loss_partial(const float* lse_img, const float* lse_txt, const float* img_rows, int ld, int label_off,
                             int R, float scale, float* out, void* stream);

/* ---- backward pass of the transformer blocks and the contrastive head (first slice).  The
 * reference ships no trainer: these differentiate the forward it defines and are pinned against autograd of the
 * imported reference (tests/golden/b32-yfcc-msclips.grads.npz).  GEMM gradients go through msclip_gemm itself
 * (dX = dY @ W with the transposed weight as W; dW = dY^T @ X with both operands transposed by
 * msclip_transpose_bf16). ---- */

/* out[c][m] = in[m][c] (bf16); columns m in [M, Mpad) are zero-filled (Mpad % 64 == 0: the K axis of a wgrad GEMM). */
int msclip_transpose_bf16(const void* in, int ldi, void* out, int ldo, int M, int C, int Mpad, void* stream);

/* The same for MANY matrices in one launch (the training step's W^T operands of the dgrad GEMMs, 48 per step): items [n_items]
 * and blk_start [n_items + 1] (blk_start[i] = first workgroup of item i = the sum of (M / 64) * ceil(C / 64) over the items
 * before it; n_blocks = blk_start[n_items]) are DEVICE arrays, built once for tensors whose storage persists.  Per item:
 * out [C][ldo] = in [M][ldi]^T, bf16; M % 64 == 0, C % 8 == 0, ldi % 8 == 0, ldo % 8 == 0, ldo >= M, 16-byte aligned bases. */
typedef struct msclip_transpose_item {
  const void* in;
  void* out;
  int ldi, ldo, M, C;
} msclip_transpose_item;
int msclip_transpose_bf16_multi(const msclip_transpose_item* items_dev, const int* blk_start_dev, int n_items, int n_blocks,
                                void* stream);

/* y = bf16(x) for an fp32 matrix (C, ldx, ldy multiples of 4): gradient streams are fp32, GEMM operands bf16. */
int msclip_cast_bf16(const float* x, int ldx, void* y, int ldy, int M, int C, void* stream);
/* msclip_cast_bf16 that also leaves the column sums of x: part [part_blocks][C] fp32 receives per-block partial sums (block b
 * sums the rows b, b + part_blocks, ...); fold them with msclip_colsum.  The training step's residual-stream gradient is both
 * the bf16 operand of its projection's dgrad / wgrad GEMMs and, summed over the tokens, that projection's bias gradient.
 * C <= 1024, C % 4 == 0.  skip_group g > 0: x holds g + 1 rows per sample and the first of each (the class token) is skipped --
 * output row m is x row m + m / g + 1 (the lateral adapters' backward: the top-down term has no class row, M.py:1768-1771). */
int msclip_cast_bf16_colsum(const float* x, int ldx, void* y, int ldy, int M, int C, float* part, int part_blocks, int skip_group,
                            void* stream);

/* out[n] (+)= sum_m x[m][n] (x bf16 or fp32): bias gradients, LayerNorm parameter gradients' second stage.  chunks > 1:
 * row chunks in parallel into scratch [chunks, N], folded in chunk order (deterministic) by the workgroup of each column block
 * that finishes last -- ONE launch (ticket counters from a per-device ring; MSCLIP_COLSUM_TWO_STAGE=1: a second launch folds);
 * chunks == 1: one launch straight into out. */
int msclip_colsum(const void* x, int ld, int is_f32, float* out, int M, int N, int accumulate, float* scratch, int chunks,
                  void* stream);

/* Many fp32 column sums in one launch per 96 items (the training step's deferred folds of per-block partial matrices: LayerNorm
 * parameter gradients, bias gradients): dst[n] = sum_m src[m][n] for n < N, rows added in a fixed order; the first scale_n results
 * are multiplied by scale.  `items` is a HOST array (copied into the kernel arguments); src / dst are device pointers. */
typedef struct msclip_fold_item {
  const float* src;
  float* dst;
  int M, N, ld;
  int scale_n;
  float scale;
} msclip_fold_item;
int msclip_colsum_multi(const msclip_fold_item* items, int n_items, void* stream);

/* QuickGELU on a saved pre-activation and its backward (M.py:222-224): y = h sigma(1.702 h);
 * dh = dy (sigma + 1.702 h sigma (1 - sigma)).  bf16, n % 8 == 0. */
int msclip_quickgelu(const void* h, void* y, long long n, void* stream);
int msclip_quickgelu_bwd(const void* h, const void* dy, void* dh, long long n, void* stream);

/* LayerNorm backward (M.py:204-219).  Row m of the op reads x[src(m)], src(m) = row_idx ? row_idx[m] : m * row_mul;
 * dy [M, C] bf16 or fp32; dx[src(m)] = (or +=) the input gradient.  part (optional) [part_blocks][2][C] receives per-block
 * partial sums of dgamma (= sum dy * xhat) and dbeta (= sum dy): fold with msclip_colsum.  C in {512, 768}.
 * dxb + sum_part (both or neither; row_idx NULL, row_mul 1): the written dx rows also leave as bf16 dxb [M][lddxb] -- in the
 * training step the residual-stream gradient behind a LayerNorm backward is the next projection's output gradient, the operand
 * of its dgrad / wgrad GEMMs -- with their per-block column sums in sum_part [part_blocks][C] (= or, with sum_accumulate, +=
 * what the previous row segment's launch left there; folded by msclip_colsum they are that projection's bias gradient). */
int msclip_layernorm_bwd(const float* x, int ldx, const int* row_idx, int row_mul, const void* dy, int lddy, int dy_is_f32,
                         const float* gamma, float* dx, int lddx, int accumulate, float* part, int part_blocks, int M,
                         int C, float eps, void* dxb, int lddxb, float* sum_part, int sum_accumulate, void* stream);

/* msclip_attention for ONE query per sample: the last block, where only the class row of an image (M.py:2685) / the EOT row
 * of a caption (M.py:3057-3060) is read afterwards.  q: bf16 [nsamples, ldqc] = the (pre-scaled) query rows; qkv: the token
 * matrix of msclip_attention -- only its k | v columns are read (the q columns may hold anything); sample b's keys are rows
 * row_base + b*L ... and their count is last_row ? last_row[b] - (row_base + b*L) + 1 : L (a causal query at its own position
 * sees the keys up to itself; NULL = all L keys).  out: bf16 [nsamples, ldo].  L <= 256; ldqc, ldq, ldo multiples of 8. */
int msclip_attention_lastq(const void* q, int ldqc, const void* qkv, int ldq, void* out, int ldo, int nsamples, int L, int heads,
                           const int* last_row, int row_base, void* stream);

/* Backward of msclip_attention for L <= 208: dqkv [q | k | v gradients] from qkv, the forward output o and its
 * gradient dout (all bf16, same layouts as the forward).  L <= 96: the whole head resident in LDS; 97-208 (the 197-token
 * grid of ViT-B/16): query axis in blocks of 32, dK / dV accumulated in registers across the blocks.
 * colsum_part (NULL: off; L <= 96 only): fp32 [nsamples][3 * heads * 64], row b = the sums over sample b's tokens of its dqkv
 * rows (fp32 values, fixed order) -- folded over the samples (msclip_colsum) they are the in_proj bias gradient, without a second
 * pass over dqkv. */
int msclip_attention_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, int nsamples, int L, int heads,
                         int ldq, int ldo, int causal, float* colsum_part, void* stream);

/* dx = (dy - y (y . dy)) / ||x||,  y = x / ||x||   (M.py:2983, :3076). */
int msclip_l2norm_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int M, int E, void* stream);

/* dL/dS of the symmetric cross-entropy on this rank's row block: G[r][j] = w (exp(S - lse_row[r]) + exp(S - lse_col[j])
 * - 2 [j == label_off + r]), S [R, N] fp32 = scale * A_loc @ B_all^T, w = 1 / (2 N_global); G bf16 [R, ldg] with columns
 * [N, Npad) zeroed; dscale_part[r] = sum_j G S (optional). */
int msclip_clip_loss_bwd_g(const float* S, int lds, const float* lse_row, const float* lse_col, int label_off, float w,
                           void* G, int ldg, float* dscale_part, int R, int N, int Npad, void* stream);

/* Token + positional embedding backward (M.py:3047-3048): dEmb[token] += dx row (fp32 atomics), dPos[l] += dx row (dpos NULL: not
 * formed -- the training step takes it as a fixed-order column sum over the batch instead, msclip_colsum). */
int msclip_embed_tokens_bwd(const long long* tokens, const float* dx, int lddx, float* demb, float* dpos, int B, int L,
                            int C, int vocab, void* stream);

/* Lateral adapter (M.py:1752-1778): the pre-LayerNorm sum [cls; BN(dw3x3(grid))] + [usecls * cls; t] (saved by the
 * training forward) and the gradient wrt the incoming tokens from the gradient of that sum. */
int msclip_adapter_sum(const float* xin, int ldx, const float* t, int ldt, const float* dww, const float* dwb, float* out,
                       int ldo, int B, int L, int g, int C, int usecls, void* stream);
int msclip_adapter_dx(const float* dsum, int lds, const float* dww, float* dx, int lddx, int B, int L, int g, int C,
                      int usecls, void* stream);

/* ---- convolutional side of the backward pass (csrc/backward_conv.hip).  The contractions run on msclip_gemm:
 * dW = dY^T . im2col(X), dX = col2im(dY . W); reference forward: M.py:1898-2000 (stem), 1812-1895 (parallel branch),
 * 1752-1778 (adapters).
 * msclip_im2col: x_kind 0 = NHWC bf16 [B, H, W, C], 1 = NCHW fp32 image, 2 = NCHW bf16 image; col bf16 [B*Ho*Wo, Kp],
 *   column (kh*KW + kw)*C + ci (the packed forward weights' K order), zero beyond KH*KW*C and outside the image.
 * msclip_col2im: dcol bf16 [B*Ho*Wo, ld] -> dx NHWC bf16 (C % 8 == 0), optionally added to what dx holds.
 * msclip_relu_bwd: out = (dy [+ dy2]) * (y > 0) over n bf16 elements (n % 8 == 0; dy2 may be null). */
int msclip_im2col(const void* x, int x_kind, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                  int Ho, int Wo, int Kp, void* stream);

/* Weight + bias gradient of a 3 x 3 / stride 2 / pad 1 convolution on the fp32 NCHW input image [B, 3, S, S] (the stem's conv1,
 * parallel stage 0: M.py:1898-1905, 1812-1830) WITHOUT a patch matrix: dy bf16 [B * Ho * Ho, lddy] (Ho = S / 2 rounded up), co
 * <= 64 channels (a multiple of 8).  part fp32 [part_blocks][co_pad = 16 ceil(co / 16)][32]: block partials of
 * dW[co][(kh * 3 + kw) * 3 + ci] (columns 0..26; the image rounded to bf16 like msclip_im2col's patch matrix), column 27 = sum
 * of dy (the bias gradient), 28..31 zero; rows co.. of a partial are zero.  Fold with msclip_colsum over [part_blocks, co_pad *
 * 32].  S % 4 == 0, S <= 256; blocks beyond the work write zeros. */
int msclip_image_conv_wgrad(const float* img, const void* dy, int lddy, float* part, int part_blocks, int B, int S, int co,
                            void* stream);
int msclip_col2im(const void* dcol, int ld, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Ho,
                  int Wo, int accumulate, void* stream);
int msclip_relu_bwd(const void* dy, const void* dy2, const void* y, void* out, long long n, void* stream);
/* Adapters' kernel == stride depthwise conv (msclip_dwpool): input gradient (dtop NHWC bf16, optionally accumulated) and
 * filter gradient as `slabs` partial sums part[slabs][k*k][C] fp32 (fold with msclip_colsum: deterministic). */
int msclip_dwpool_bwd(const void* dpool, int ldp, const float* w, void* dtop, int B, int H, int W, int C, int k,
                      int accumulate, void* stream);
int msclip_dwpool_wgrad(const void* dpool, int ldp, const void* top, float* part, int B, int H, int W, int C, int k,
                        int slabs, void* stream);
/* Filter gradient of the depthwise 3x3 over the token grid (msclip_adapter_sum's dww): part[slabs][9][C] fp32. */
int msclip_dw3x3_wgrad(const float* dsum, int lds, const float* x, int ldx, float* part, int B, int L, int g, int C,
                       int slabs, void* stream);

/* ---- train-mode BatchNorm on a raw convolution output x [M, C] (per-GPU batch statistics; reference nn.BatchNorm2d in
 * train(), M.py:1825-1861, 1920-1936).  x_f32 / dy_f32 / y_f32: 0 = bf16, 1 = fp32 matrices (dx has dy's type).
 * msclip_bn_stats: part[chunks][2][C] = per-row-chunk (sum x, sum x^2); fold with msclip_colsum.
 * msclip_bn_apply: y = act(x * scale[c] + shift[c] [+ resid (bf16)]).
 * msclip_bn_bwd_reduce: part[chunks][2][C] = (sum dy, sum dy * xhat), xhat = (x - mean) * rstd.
 * msclip_bn_bwd_dx: dx = gamma * rstd * (dy - dbeta / n_stat - xhat * dgamma / n_stat), n_stat = rows the statistics cover
 *   (= M unless the caller passes r consecutive rows as one row of r*C columns with r-fold repeated channel vectors). */
int msclip_bn_stats(const void* x, int ld, int x_f32, float* part, int M, int C, int chunks, void* stream);
int msclip_bn_apply(const void* x, int ld, int x_f32, const float* scale, const float* shift, const void* resid, int ldr,
                    void* y, int ldy, int y_f32, int M, int C, int relu, void* stream);
/* msclip_bn_finish with every output vector written `rep` times back to back (out [5][rep * C]): the tiled per-channel vectors the
 * passes over r-folded narrow maps read.  msclip_bn_bwd_finish: part [chunks][2][r * C] of msclip_bn_bwd_reduce ->
 * out [3][r * C] = (dbeta, dgamma, gamma), each tiled r times (fixed summation order). */
int msclip_bn_finish_tiled(const float* sums, int r, int C, long long n, const float* gamma, const float* beta, float eps,
                           float* out, int rep, void* stream);
int msclip_bn_bwd_finish(const float* part, int chunks, int r, int C, const float* gamma, float* out, void* stream);
int msclip_bn_bwd_reduce(const void* dy, int lddy, int dy_f32, const void* x, int ld, int x_f32, const float* mean,
                         const float* rstd, float* part, int M, int C, int chunks, void* stream);
int msclip_bn_bwd_dx(const void* dy, int lddy, int dy_f32, const void* x, int ld, int x_f32, const float* mean,
                     const float* rstd, const float* gamma, const float* dbeta, const float* dgamma, void* dx, int lddx, int M,
                     int C, long long n_stat, void* stream);
/* msclip_bn_bwd_fused (round 6): the two passes above for the shape the training step has (bf16 gradients, fp32 raw maps, C and
 * every leading dimension % 4 == 0, 16-byte aligned bases), four columns per thread, with
 *   - the ReLU in front of the BatchNorm applied on the fly: d = bf16(dy [+ dy2]) * (y > 0), bit for bit what msclip_relu_bwd
 *     writes (y = the block's bf16 output; dy2 / y may be null), and
 *   - one or two BatchNorms behind the same d (a residual block's main path and its shortcut: M.py:1898-1936, 1812-1861) per pass.
 * pass 0: s->part [chunks][2][C] = (sum d, sum d * xhat) per row chunk (fold with msclip_bn_bwd_finish);
 * pass 1: s->dx (bf16) = gamma rstd (d - dbeta / n_stat - xhat dgamma / n_stat).  Not recordable in a plan (host structs). */
typedef struct msclip_bn_bwd_side {
  const void* x;        /* [M][ld]: the raw convolution output, fp32 -- or (x_bf16) xhat in bf16 with mean = 0, rstd = 1, gamma := gamma rstd */
  int ld;
  const float* mean;    /* [C] */
  const float* rstd;    /* [C] */
  const float* gamma;   /* pass 1 */
  const float* dbeta;   /* pass 1 */
  const float* dgamma;  /* pass 1 */
  void* dx;             /* pass 1: bf16 [M][lddx] */
  int lddx;
  float* part;          /* pass 0 */
  int x_bf16;           /* both sides of a call alike */
} msclip_bn_bwd_side;
int msclip_bn_bwd_fused(int pass, const void* dy, int lddy, const void* dy2, int lddy2, const void* y, int ldy,
                        const msclip_bn_bwd_side* s1, const msclip_bn_bwd_side* s2, int M, int C, int chunks, long long n_stat,
                        void* stream);

/* One parameter tensor of msclip_adamw_multi (host-side array; lr / weight_decay per tensor = the reference's parameter
 * groups, lib/optim/build.py via CUSTOM.LR_SHARE / WD_SHARE and TRAIN.WITHOUT_WD_LIST). */
typedef struct msclip_adamw_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
  float lr;
  float weight_decay;
  void* pk;        /* optional: a packed copy of the updated values, pk[i] = p[i] * pk_scale (the engine's bf16 operand of a */
  float pk_scale;  /* projection weight, the 64^-0.5 of the q rows folded in), written by the same kernel; NULL = none      */
  int pk_f32;      /* element type of pk: 0 = bf16, 1 = fp32                                                                */
} msclip_adamw_tensor;

/* AdamW with decoupled weight decay on one fp32 tensor (step >= 1 for the bias corrections). */
int msclip_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, void* stream);
/* The same update for `count` tensors in a handful of launches (32 K-element chunks of up to 36 tensors per launch, the
 * tensor table travels in the kernel arguments): bitwise the result of `count` msclip_adamw calls.  `tensors` is a HOST
 * array, read before the call returns. */
int msclip_adamw_multi(const msclip_adamw_tensor* tensors, int count, float beta1, float beta2, float eps, int step,
                       void* stream);

/* ---- Re-packing the convolutional side's derived weights after an optimizer step in ONE launch (msclip_amd/packing.py's folds:
 * eval-mode BatchNorm into the filter, M.py:1825-1861 / 1920-1936; the stem stages' 1x1 shortcut into the 3x3 centre tap; layouts).
 * One item per derived tensor, the table and the per-item first-block table live in DEVICE memory (built once: sources are the
 * module's parameter storage, destinations the engine's persistent operands).  BatchNorm 1 = (g, b, mu, var, eps) over the
 * output channels: scale = g / sqrt(var + eps), shift = b - mu scale (IEEE fp32, no contraction).
 *   out, mode 0: bf16 [co][kpad], k = (kh, kw, ci), zero padded: w scale (+ w2 [co][ci] scale2 at the centre tap);
 *        mode 1: fp32 transposed out[k * ld + col0 + o], k = (ci, kh, kw): w scale;          out NULL: no weight output;
 *   bias_out[bias_col0 + o], bias_mode 1: shift; 2: shift + shift2 (BatchNorm 2); 3: sum_c w[o][c] shift[c] with BatchNorm 1
 *   over the INPUT channels (a pointwise conv behind a BatchNorm); 0 / bias_out NULL: none.
 * blk_start [n_items + 1]: first workgroup of every item (co * ceil(row length / 1024) workgroups, row length = kpad (mode 0) or
 * ci kh kw (mode 1); at least 1), ascending. */
typedef struct msclip_pack_item {
  const float* w;
  const float *g, *b, *mu, *var;
  const float* w2;
  const float *g2, *b2, *mu2, *var2;
  void* out;
  float* bias_out;
  float eps, eps2;
  int co, ci, kh, kw, kpad, mode, col0, ld, bias_mode, bias_col0;
} msclip_pack_item;
int msclip_pack_weights(const msclip_pack_item* items_dev, const int* blk_start_dev, int n_items, int n_blocks, void* stream);

/* HIP streams with an explicit priority on the current device.  The training step runs its weight-gradient jobs beside the
 * dgrad chain (the role torch DDP's / autograd's side streams play under the reference's lib/core/function.py:66-77
 * backward); streams of the LOWEST priority draw their hardware queue from a pool of their own, so the overlap does not
 * depend on which other streams (RCCL's, the caller's) the runtime happens to map onto the compute stream's queue.
 * priority_range: numerically greatest = least urgent (HIP's convention).  Return value: hipError_t. */
int msclip_stream_priority_range(int* least, int* greatest);
int msclip_stream_create(int priority, void** stream);
int msclip_stream_destroy(void* stream);

/* Library / device introspection (no GPU work). */
/* Chain rule of a frozen-statistics BatchNorm folded into its convolution (W_f = W gamma rstd, shift = beta - mean gamma rstd)
 * back to the module's parameters: dW = G s, dgamma = (sum_k G W - mean dshift) rstd, dbeta = dshift; G = dL/dW_f [cout, ldg >= K]
 * fp32, w_raw [cout, K] the raw filter, one block per output channel (M.py:1825-1861, 1920-1936 differentiated in eval() mode). */
int msclip_bn_fold_bwd(const float* G, long long ldg, const float* w_raw, int cout, int K, const float* dshift, const float* gamma,
                       const float* mean, const float* var, float eps, float* dW, float* dgamma, float* dbeta, void* stream);

/* Train-mode BatchNorm: per-channel tail of the statistics pass.  sums [2][r][C] = (sum x, sum x^2) per row fold (msclip_bn_stats
 * partials, folded by msclip_colsum) over n values per channel -> out [5][C] = mean, biased variance, 1/sqrt(var + eps),
 * scale = gamma rstd, shift = beta - mean scale (what msclip_bn_apply and the backward read). */
int msclip_bn_finish(const float* sums, int r, int C, long long n, const float* gamma, const float* beta, float eps, float* out,
                     void* stream);

/* Allocates what the library would otherwise allocate lazily on the current device (the column sums' ticket counters): call once
 * per device before capturing a stream or recording a launch plan that contains training-step launches. */
int msclip_prepare_device(void);

/* ---- Launch plans (round 6; no reference counterpart: the reference's step is a Python loop over ATen calls, M.py:2388-2459,
 * 3126-3141).  A plan is the launch table of one step: while a plan is recording on the calling thread every stream-ordered entry
 * point of this library appends (itself, a copy of its arguments, the slot of its stream) before doing its work -- the recording
 * pass is a real step -- and msclip_plan_run replays the table: the same entry points with the stored arguments on the streams
 * the caller passes, no host-side argument marshalling.  Data-dependent row counts must be device-side (M_dev / m_dev / dims).
 *   msclip_plan_begin(plan, streams, n): start recording; streams[i] is slot i (slot 0 = the caller's main stream).  A launch on
 *     any other stream makes msclip_plan_end fail.  One recording per thread at a time.
 *   msclip_plan_bind_external(plan, base, nbytes) -> index: pointer arguments inside [base, base + nbytes) (inputs that live in
 *     caller buffers: images, token ids) are re-based at run time on ext[index] of msclip_plan_run.
 *   msclip_plan_record_event(plan, stream) -> event id; msclip_plan_wait_event(plan, stream, id): the cross-stream edges of the
 *     step (the caller performs the same record / wait on its own events for the recording pass itself).
 *   msclip_plan_end -> number of table entries (< 0: unusable).  msclip_plan_abort: give up a recording.
 *   msclip_plan_run(plan, streams, n, ext, n_ext): replay.  Legal under a stream capture whose origin stream is slot 0 provided
 *     every other slot is first touched by a recorded wait on an event of a captured stream (the engine's schedules are).
 *   Entry points that read HOST arrays (msclip_colsum_multi, msclip_adamw_multi) cannot be recorded: the recording fails. */
typedef struct msclip_plan_s msclip_plan_s;
int msclip_plan_create(msclip_plan_s** plan);
int msclip_plan_destroy(msclip_plan_s* plan);
int msclip_plan_begin(msclip_plan_s* plan, void* const* streams, int nstreams);
int msclip_plan_bind_external(msclip_plan_s* plan, const void* base, long long nbytes);
int msclip_plan_record_event(msclip_plan_s* plan, void* stream);
int msclip_plan_wait_event(msclip_plan_s* plan, void* stream, int event);
int msclip_plan_end(msclip_plan_s* plan);
int msclip_plan_abort(msclip_plan_s* plan);
int msclip_plan_info(const msclip_plan_s* plan, int* n_ops, int* n_launches, int* n_events, int* n_streams, int* n_ext);
const char* msclip_plan_op_name(const msclip_plan_s* plan, int i);
int msclip_plan_run(msclip_plan_s* plan, void* const* streams, int nstreams, const void* const* ext, int next);
int msclip_plan_size(const msclip_plan_s* plan);                     /* entries so far (also while recording) */
/* Launch probes: HIP timing events around table entries op_idx[0..n) on each entry's own stream for the next `runs` replays (the
 * per-kernel durations of bench.py's roofline leg, measured inside the timed region); elapsed(run, i) after a synchronise. */
int msclip_plan_probe_enable(msclip_plan_s* plan, const int* op_idx, int n, int runs);
int msclip_plan_probe_disable(msclip_plan_s* plan);
int msclip_plan_probe_runs(const msclip_plan_s* plan);
int msclip_plan_probe_elapsed(msclip_plan_s* plan, int run, int i, float* ms);

/* A non-blocking stream whose kernels may only run on n_cus compute units (hipExtStreamCreateWithCUMask), the first n_cus bits of
 * the mask (from_top = 0) or the last (1); gfx950 enumerates CUs XCD-interleaved, so either is an even spread over the XCDs. */
int msclip_stream_create_cu_masked(int n_cus, int from_top, void** stream);

/* ---- RCCL behind the C ABI.  Replaces torch.distributed's all_gather of reference lib/utils/comm.py:140-154
 * (called from M.py:3138-3141) and the all-reduce of lib/utils/utils.py:66-73 with stream-ordered calls on the caller's stream.
 *   msclip_comm_unique_id(id128): 128 bytes naming a new communicator (one rank creates, the host distributes them);
 *   msclip_comm_init(rank, world, id128, &comm): collective; msclip_comm_destroy; msclip_comm_async_error (0 = healthy);
 *   msclip_allgather_feats: recv[r][count] = rank r's send[count] (rank-major, comm.py:150-153); dtype 0 bf16, 1 fp32, 2 bytes, 3 int32;
 *   msclip_allreduce: element-wise sum (op 0) / max (op 1) over the ranks. */
int msclip_comm_unique_id(void* id128);
int msclip_comm_init(int rank, int world, const void* id128, void** comm);
int msclip_comm_destroy(void* comm);
int msclip_comm_async_error(void* comm);
int msclip_allgather_feats(void* comm, const void* send, void* recv, long long count, int dtype, void* stream);
int msclip_allreduce(void* comm, const void* send, void* recv, long long count, int dtype, int op, void* stream);

#define MSCLIP_ABI_VERSION 8   /* 8 (round 6, late): msclip_gemm_desc.bn_mode / part_rows / bn_consts, msclip_bn_bwd_fused, msclip_stem_conv3x3s2_dual_stats / _norm; 7 (round 6): device-side row counts (M_dev / m_dev / dims), the plan executor, RCCL entry points, msclip_prepare_device; 5-6 (round 5): packed-caption entry points, msclip_qkv_attention / msclip_qkvattn_tables, msclip_pack_weights, single-launch msclip_colsum */
int msclip_abi_version(void);
const char* msclip_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif
