"""Fold what scripts/collect_evidence_{a,b,c}.sh left under gpurun_out/<tag>_{a,b,c}/ into profiles/<round>_*: wave / service timing
(diagnostic builds, each naming its own code and the product's), the all-rays parity logs, the bench lines, the five-rank
rehearsal, the random sweeps, the persistent-wave mode check.  (scripts/summarize_profiles.py and summarize_isa_budget.py do the
rocprofv3 side.)   usage: python scripts/summarize_evidence.py <tag> <round>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
A, B, C = (os.path.join(G, f"{tag}_{k}") for k in "abc")
if os.path.exists(os.path.join(A, "shas.json")):
    shas = json.load(open(os.path.join(A, "shas.json")))
    out = {"_note": "scripts/wave_times.py with the -DPGR_WAVE_TIMES build of the round's final sources (scripts/build_variants.py wavetimes; its own "
                    "device_code_sha256 below -- the stamps cost the kernel several per cent): s_memrealtime at the start and end of every 64-ray packet "
                    "+ HW_ID.  Last of three passes each.",
           "device_code_sha256_of_the_instrumented_build": shas["wavetimes"], "device_code_sha256_of_the_product_it_was_built_beside": shas["product"]}
    for k, f in (("rays_1e6_static_deal_of_whole_workgroups", "wave_times_static_1e6.json"), ("rays_1e6_persistent_waves", "wave_times_persistent_1e6.json"),
                 ("headline_1e5_trajectories", "wave_times_1e5_trajectories.json")):
        out[k] = json.load(open(os.path.join(A, f)))[-1]
    json.dump(out, open(os.path.join(P, f"{rnd}_wave_times.json"), "w"), indent=1)
    st = json.load(open(os.path.join(A, "service_times.json")))
    st["_note"] = ("scripts/service_times.py with the -DPGR_SVC_TIMING build of the round's final sources (scripts/build_variants.py svctiming): s_memtime "
                   "stamps between the sections of the bounce SERVICE phase, the fan's 64 steepest rays alone on the chip; cycles per service (stamps "
                   "included, ~40 cycles each)")
    st["device_code_sha256_of_the_instrumented_build"] = shas["svctiming"]
    st["device_code_sha256_of_the_product_it_was_built_beside"] = shas["product"]
    json.dump(st, open(os.path.join(P, f"{rnd}_service_times.json"), "w"), indent=1)
    pc = json.load(open(os.path.join(A, "persist_check.json")))
    json.dump({"_note": "scripts/persist_check.py: PGR_OPT_PERSISTENT 0 (static deal of whole workgroups) / 1 (default) / 2 (every packet from the "
                        "list's head) / 3 (the SIMD partners' first packets from its cheap end): every output array of every ray the same bits, kernel ms of each",
               "device_code_sha256": shas["product"], "cases": pc}, open(os.path.join(P, f"{rnd}_persist_check.json"), "w"), indent=1)
    print("wave / service timing, persist_check")
for f in ("bitparity_S1001.txt", "bitparity_1e6_rays.txt"):
    if os.path.exists(os.path.join(B, f)):
        shutil.copy(os.path.join(B, f), os.path.join(P, f"{rnd}_{f}")); print(f)
for f in ("bench_line.json", "bench_line_config2.json", "bench_line_config2_blocked.json", "bench_line_flatearth.json", "bench_line_rays_1e6.json",
          "bench_line_5ranks_one_gpu_rehearsal.json", "fuzz_sweeps.txt"):
    if os.path.exists(os.path.join(C, f)) and os.path.getsize(os.path.join(C, f)) > 0:
        shutil.copy(os.path.join(C, f), os.path.join(P, f"{rnd}_{f}")); print(f)
