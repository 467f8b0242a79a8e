"""Variant builds of the library for A/B experiments (scripts/kbench.py --lib scripts/ab/<name>.so)."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pygenray_amd import _lib
VARIANTS = {"pow2ulp": ["-DPGR_POW_2ULP"], "noreplay": ["-DPGR_NO_REPLAY"], "libmtrig": ["-DPGR_LIBM_TRIG"],
            "dbgreplay": ["-DPGR_DBG_REPLAY"], "nobandtab": ["-DPGR_NO_BAND_TABLE"],
            "plain": [],
            # round 3: the sample-store experiments and the two-step Newton of the controller's power
            "ring": ["-DPGR_SAMPLE_RING=1"], "defer": ["-DPGR_DEFER_STORES=1"], "wavering": ["-DPGR_WAVE_RING=1"],
            "st1": ["-DPGR_STORE_EXPERIMENT=1"], "st2": ["-DPGR_STORE_EXPERIMENT=2"], "pow2n": ["-DPGR_POW_TWO_NEWTON"],
            "timing": ["-DPGR_TIMING"], "keepk0": ["-DPGR_KEEP_K=0"], "smptrips": ["-DPGR_DBG_SAMPLE_TRIPS"], "pinlit": ["-DPGR_PIN_LITERALS=1"], "pinlit_nop": ["-DPGR_PIN_LITERALS=1", "-DPGR_PIN_P=0"], "nopin_p": ["-DPGR_PIN_P=0"]}
for name in (sys.argv[1:] or VARIANTS):
    print(name, _lib.build(force=True, out=_lib.CSRC + f"/../../scripts/ab/{name}.so", extra_flags=VARIANTS[name], verbose=True), flush=True)
