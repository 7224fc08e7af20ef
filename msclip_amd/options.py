"""Run-time options of the engine in ONE object (round 6: replaces call-time environment reads inside the engine).

`EngineOptions.from_env()` is read ONCE, when an Engine is built (model.engine()); afterwards a stale shell variable cannot
change numerics or schedules half-way through a run, and a test sets what it wants explicitly:

    eng = model.engine()
    eng.opt = eng.opt.replace(plan=False)          # e.g. the eager launch loop for an A/B

Every field names the environment variable it is initialised from (0 / 1), so existing command lines keep working.  Switches of
retired A/B paths whose result is on record in DESIGN.md are not here any more.
"""
import dataclasses
import os


def _flag(name, default):
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    return v != "0"


@dataclasses.dataclass(frozen=True)
class EngineOptions:
    # ---- text rows
    text_pack: bool = True            # MSCLIP_TEXT_PACK: captions own argmax + 1 rows instead of context_length
    dynamic_rows: bool = True         # MSCLIP_DYNAMIC_ROWS: the packed row count stays on the device (msclip_text_lengths dims ->
    #                                   M_dev / m_dev): no host read per batch, the step is capturable / plannable.  0 = the
    #                                   round-5 path (host reads the total, sizes every launch exactly)
    # ---- launch loop
    plan: bool = True                 # MSCLIP_PLAN: record the step once into a native launch table (msclip_plan_*), replay it per call
    # ---- schedule
    conv_side_stream: bool = True     # MSCLIP_CONV_SIDE_STREAM: parallel conv branch + adapter tops on a side stream
    text0_stream: bool = True         # MSCLIP_TEXT0_STREAM: text front + text block 0 beside the image front
    branch_early: bool = True         # MSCLIP_BRANCH_EARLY: the branch starts behind the fused stem kernel, not the whole front
    side_priority: str = "normal"     # MSCLIP_SIDE_PRIORITY: "low" = the conv side stream is created with the lowest HIP stream priority
    side_cu_mask: int = 0             # MSCLIP_SIDE_CUS: > 0 = the conv side stream is created with a CU mask of this many CUs
    #                                   (whole XCD-interleaved set; hipExtStreamCreateWithCUMask), 0 = an ordinary stream
    # ---- kernel / path selection kept as options (each has a test or a documented A/B that flips it)
    ln_fold: bool = True              # MSCLIP_LN_FOLD
    fused_qkv_attn: bool = False      # MSCLIP_FUSED_QKV_ATTN (north_star's fused QKV + SDPA kernel; measured 0.6 % slower: opt-in)
    front_unfused: bool = False       # MSCLIP_FRONT_UNFUSED
    block_unfused: bool = False       # MSCLIP_BLOCK_UNFUSED
    full_last_block: bool = False     # MSCLIP_FULL_LAST_BLOCK
    last_block_all_queries: bool = False   # MSCLIP_LAST_BLOCK_ALL_QUERIES
    adapter_ln1_pass: bool = False    # MSCLIP_ADAPTER_LN1_PASS
    gather_packed: bool = False       # MSCLIP_GATHER_PACKED: one [B, 2, E] all-gather instead of one per modality
    repack_table: bool = True         # MSCLIP_REPACK_TABLE
    native_collectives: bool = False  # MSCLIP_NATIVE_COLLECTIVES: feature gather / loss all-reduce through the C ABI's RCCL entry
    #                                   points (msclip_comm_*) on the compute stream instead of ProcessGroupNCCL

    @classmethod
    def from_env(cls):
        return cls(
            text_pack=_flag("MSCLIP_TEXT_PACK", True),
            dynamic_rows=_flag("MSCLIP_DYNAMIC_ROWS", True),
            plan=_flag("MSCLIP_PLAN", True),
            conv_side_stream=_flag("MSCLIP_CONV_SIDE_STREAM", True),
            text0_stream=_flag("MSCLIP_TEXT0_STREAM", True),
            branch_early=_flag("MSCLIP_BRANCH_EARLY", True),
            side_priority=os.environ.get("MSCLIP_SIDE_PRIORITY", "normal"),
            side_cu_mask=int(os.environ.get("MSCLIP_SIDE_CUS", "0") or 0),
            ln_fold=_flag("MSCLIP_LN_FOLD", True),
            fused_qkv_attn=_flag("MSCLIP_FUSED_QKV_ATTN", False),
            front_unfused=_flag("MSCLIP_FRONT_UNFUSED", False),
            block_unfused=_flag("MSCLIP_BLOCK_UNFUSED", False),
            full_last_block=_flag("MSCLIP_FULL_LAST_BLOCK", False),
            last_block_all_queries=_flag("MSCLIP_LAST_BLOCK_ALL_QUERIES", False),
            adapter_ln1_pass=_flag("MSCLIP_ADAPTER_LN1_PASS", False),
            gather_packed=_flag("MSCLIP_GATHER_PACKED", False),
            repack_table=_flag("MSCLIP_REPACK_TABLE", True),
            native_collectives=_flag("MSCLIP_NATIVE_COLLECTIVES", False),
        )

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


@dataclasses.dataclass(frozen=True)
class TrainOptions:
    """The training step's remaining switches (msclip_amd/train.py, train_conv.py, gradgemm.py), read from the environment ONCE at
    import; tests / probes swap the object: `options.TRAIN = options.TRAIN.replace(dgrad_col2im=True)`.  Round 6 removed the
    switches of retired A/B paths whose numbers are on record in DESIGN.md s8 (MSCLIP_FOLDS_EAGER, _WT_PER_MATRIX, _LN_BWD_UNFUSED,
    _BIAS_COLSUM_PASS, _IMAGE_COLS_64, _DGRAD_PARITY4, _SHORTCUT_COL2IM, _RELU_BWD_PASS, _FIRST_CONV_GEMM, _ADAPTER_BWD_UNFUSED,
    _WGRAD_TN, _WGRAD_TN_RAGGED): the shipped path is the only one now, the general-shape fallbacks they guarded stay."""
    wgrad_sync: bool = False          # MSCLIP_WGRAD_SYNC: weight-gradient jobs on the calling stream instead of the lane stream
    dgrad_col2im: bool = False        # MSCLIP_DGRAD_COL2IM: conv input gradients through column matrices + col2im (cross-check of the
    #                                   parity-class implicit GEMMs, tests/test_gpu_train.py)
    im2col_main: bool = False         # MSCLIP_IM2COL_MAIN: the conv side's column matrices on the main stream instead of the lane
    compact_last_block: bool = True   # MSCLIP_TRAIN_COMPACT_LAST: the last block's out_proj / ln_2 / MLP (forward and backward) on the
    #                                   Bi + Bt rows that are read behind it (cls / EOT), as the inference path does (round 6)
    bn_bwd_fused: bool = True         # MSCLIP_BN_BWD_FUSED: train-mode BatchNorm backward with the ReLU mask applied on the fly and the
    #                                   two BatchNorms of a residual block in one pass (msclip_bn_bwd_fused); 0 = msclip_relu_bwd
    #                                   + one msclip_bn_bwd_reduce / _dx pair per BatchNorm (round 5)
    bn_two_pass: bool = True          # MSCLIP_BN_TWO_PASS: train-mode BatchNorm of the two convolutions on the input image in two passes
    #                                   over the image (statistics, then normalise + ReLU in the epilogue; the normalised values
    #                                   kept in bf16 for the backward) instead of raw fp32 maps + a statistics + a normalise pass
    raw_pack_table: bool = True       # MSCLIP_RAW_PACK_TABLE: train-mode BatchNorm's raw conv operands refreshed by one msclip_pack_weights
    #                                   launch per step instead of ~60 ATen launches
    adapter_bn_views: bool = True     # MSCLIP_ADAPTER_BN_VIEWS: the lateral adapters' batch-statistics BatchNorm (over the grid rows of a
    #                                   token matrix) on whole samples viewed as rows of L * D columns, the class token's columns
    #                                   neutralised -- no gather / clone / scatter copies of the [B L, D] maps (round 6)
    colsum_main: bool = False         # MSCLIP_COLSUM_MAIN: the conv side's bias sums on the main stream instead of the lane
    #                                   (either of the two makes a hipGraph replay of the step right: profiles/r06_train_hipgraph_probe.txt)

    @classmethod
    def from_env(cls):
        return cls(wgrad_sync=_flag("MSCLIP_WGRAD_SYNC", False), dgrad_col2im=_flag("MSCLIP_DGRAD_COL2IM", False),
                   im2col_main=_flag("MSCLIP_IM2COL_MAIN", False), colsum_main=_flag("MSCLIP_COLSUM_MAIN", False),
                   compact_last_block=_flag("MSCLIP_TRAIN_COMPACT_LAST", True), bn_bwd_fused=_flag("MSCLIP_BN_BWD_FUSED", True),
                   bn_two_pass=_flag("MSCLIP_BN_TWO_PASS", True), raw_pack_table=_flag("MSCLIP_RAW_PACK_TABLE", True),
                   adapter_bn_views=_flag("MSCLIP_ADAPTER_BN_VIEWS", True))

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


TRAIN = TrainOptions.from_env()
