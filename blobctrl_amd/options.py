"""Planner options of the denoise plans (round 6: VERDICT r5 item 7, "quarantine the switches").

Rounds 1-5 grew 51 `BC_*` environment variables, each read where its experiment happened to sit.  What is left:

  * ONE variable for everything the PLANNER decides, `BC_PLAN="key=value,key=value"`, parsed here against the table below (unknown keys and
    malformed values raise - a typo must not silently run the default plan).  Every key either switches a fused / specialised kernel back
    to the launch list it replaced (results equal up to fp16 rounding: the block and loop fixtures run in these modes too) or moves a
    planning constant whose measured optimum is the default.  No key changes what is computed: the two diagnostics that produced WRONG
    results by design (weights aliased to one storage; ablation bits of the halo convolution) are gone from the package - the halo
    kernel's bits now need a `-DBC_DIAGNOSTICS` build of the library.
  * a handful of run-time switches of the loop driver / library (`BC_NO_GRAPHS`, `BC_LOOP_GRAPH`, `BC_ONE_STREAM`, `BC_NO_MODULE_GRAPHS`,
    `BC_SPLIT_CFG`, `BC_NO_TUNING`, `BC_TUNING_FILE`, `BC_GN_UNFUSED`, `BC_GEMM_GENERIC`, `BC_GEMM_TILE`, `BC_ATTN_NO8`) and the in-kernel cycle
    stamps (`BC_WREG_STAMPS`, `BC_RC_STAMPS`), which synchronise the stream but leave results alone.

`effective()` is recorded into every plan (`Plan.options`), and `bench.py` prints the non-default entries plus the count of launches per
kernel variant that actually ran (`config.plan_variants`): a bench line states which plan it measured.
"""
import os

# key -> (default, meaning).  bool keys take 0 / 1.
TABLE = {
    # ---- fused / specialised kernels (0 = the launch list they replaced)
    "rowchain": (True, "transformer blocks of 320 / 640 channels on csrc/rowchain.hip (0: GroupNorm / LayerNorm / GEMM launch list)"),
    "midx": (True, "cross-attention of the 320 / 640-channel blocks inside the row-chain's MID launch (0: MID + bc_attention)"),
    "ffp": (True, "UNet's split 640-channel block end as OUT_FFP + sum (0: OUT_FF + OUT_TAIL)"),
    "ffp_blob": (False, "BlobNet's split block end as OUT_FFP + sum too"),
    "gw": (True, "1280-channel projections at M <= gw_maxm on csrc/gemm_wreg.hip (0: LDS-DMA tiles + LayerNorm launches)"),
    "ctx_fold": (True, "cross-attention of the 1280-channel blocks folded over the prompt (0: to_q + bc_attention + to_out)"),
    "g256": (True, "large-M dense projections on csrc/gemm256.hip from g256_min_tiles tiles on (0: gemm_fast tiles)"),
    "gw_ff1_g256": (True, "ff.net.0 of the 1280-channel blocks at M <= gw_maxm as LayerNorm + gemm256.hip where it has >= g256_min_tiles tiles"),
    "cfg_prefix": (True, "CFG-invariant prefix of the UNet once per image pair (0: every launch at the full CFG batch)"),
    "wreg": (True, "ResBlock convolutions on csrc/conv_wreg.hip (0: csrc/conv_halo.hip)"),
    "halo": (True, "ResBlock convolutions with the fused GroupNorm prologue at all (0: GroupNorm pass + implicit GEMM)"),
    "gn_finalize_launch": (False, "a bc_gn_finalize launch per GroupNorm instead of the finalize in the consumer's prologue"),
    # ---- planning constants (defaults = the measured optima: DESIGN 3.6, 3.7, 6)
    "halo_ctas": (0, "workgroups a split-K convolution pass aims for (0: 128 on conv_wreg, 256 on conv_halo)"),
    "halo_min_cps": (0, "fewest 64-channel chunks per workgroup of a split pass (0: 3 on conv_wreg, 2 on conv_halo)"),
    "halo_full": (0, "workgroups from which a pass runs unsplit (0: two thirds of halo_ctas)"),
    "rowchain_min_blocks_640": (64, "fewest 64-row blocks for the 640-channel row-chain"),
    "rowchain_min_blocks_640_blob_up": (32, "... for BlobNet's up blocks"),
    "ff_split_640": (0, "workgroups per row block of the 640-channel block end up to 64 row blocks (0: 4 with OUT_FFP, 2 without)"),
    "ff_split_640_blob": (0, "... for BlobNet (0: as ff_split_640)"),
    "ff_split_320": (1, "workgroups per row block of the 320-channel block end up to 128 row blocks"),
    "gw_maxm": (1024, "largest M of a projection on gemm_wreg.hip"),
    "ctx_fold_maxb": (2, "largest UNet batch that takes the prompt-folded cross-attention"),
    "gn_pass_min_requests": (4, "edit requests per batch from which a ResBlock convolution runs as GroupNorm pass + plain conv_wreg (0: never)"),
    "gn_pass_min_cin": (320, "... and fewest input channels of such a convolution"),
    "g256_min_tiles": (64, "fewest 256 x 256 tiles for gemm256.hip (a quarter of the CUs: the other queue has the rest)"),
}


def _parse(text):
    out = {}
    for item in (text or "").replace(";", ",").split(","):
        item = item.strip()
        if not item:
            continue
        if "=" not in item:
            raise ValueError(f"BC_PLAN: '{item}' is not key=value")
        k, v = (s.strip() for s in item.split("=", 1))
        if k not in TABLE:
            raise ValueError(f"BC_PLAN: unknown key '{k}' (known: {', '.join(sorted(TABLE))})")
        try:
            iv = int(v)
        except ValueError:
            raise ValueError(f"BC_PLAN: {k}={v} is not an integer") from None
        out[k] = bool(iv) if isinstance(TABLE[k][0], bool) else iv
    return out


def effective():
    """Every option with its value for plans recorded NOW (BC_PLAN is read when a plan is recorded, not at import)."""
    vals = {k: d for k, (d, _) in TABLE.items()}
    vals.update(_parse(os.environ.get("BC_PLAN")))
    return vals


def opt(key):
    if key not in TABLE:
        raise KeyError(key)
    return effective()[key]


def non_default():
    return {k: v for k, v in effective().items() if v != TABLE[k][0]}
