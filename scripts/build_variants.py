"""Variant builds of the library for A/B experiments (scripts/kbench.py --lib scripts/ab/<name>.so)."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pygenray_amd import _lib
VARIANTS = {"pow2ulp": ["-DPGR_POW_2ULP"], "noreplay": ["-DPGR_NO_REPLAY"], "libmtrig": ["-DPGR_LIBM_TRIG"],
            "nobandtab": ["-DPGR_NO_BAND_TABLE"], "plain": [],
            "pow2n": ["-DPGR_POW_TWO_NEWTON"], "noziv": ["-DPGR_POW_NO_ZIV"], "timing": ["-DPGR_TIMING"], "keepk0": ["-DPGR_KEEP_K=0"],
            "pinlit": ["-DPGR_PIN_LITERALS=1"], "pinlit_nop": ["-DPGR_PIN_LITERALS=1", "-DPGR_PIN_P=0"], "nopin_p": ["-DPGR_PIN_P=0"],
            "wavetimes": ["-DPGR_WAVE_TIMES"], "svctiming": ["-DPGR_SVC_TIMING"],
            "cellrec": ["-DPGR_CELL_RECORDS"], "rowpairs": ["-DPGR_ROW_PAIRS"]}
# (the round-3 sample-store and service-timing switches -- PGR_SAMPLE_RING, PGR_WAVE_RING, PGR_DEFER_STORES,
# PGR_STORE_EXPERIMENT, PGR_DBG_REPLAY, PGR_DBG_SAMPLE_TRIPS -- left the kernel with
# scripts/experiments/r03_sample_store_experiments.patch; it applies to commit c081c9f (a worktree of that commit builds them again))
for name in (sys.argv[1:] or VARIANTS):
    print(name, _lib.build(force=True, out=_lib.CSRC + f"/../../scripts/ab/{name}.so", extra_flags=VARIANTS[name], verbose=True), flush=True)
