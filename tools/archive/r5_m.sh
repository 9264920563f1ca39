cd $GRAFT_REPO_ROOT
timeout -k 5 900 python - <<PY
import json, time, bench
for name in ("cfg3_main_4k_10b_ra", "cfg4_main_8k_10b_ra"):
    wl = bench.WORKLOADS[name]
    t = time.time()
    rd = bench.reference_decoder_leg(wl)
    print(name, round(time.time() - t, 1), "s")
    print("  plain", {k: v.get("decode_only_fps") for k, v in rd["evc_decode_on_gpu"].items()}, rd["bit_exact"])
    print("  refine", json.dumps(rd["evc_decode_dmvr_hmvp_mmvd"])[:900])
PY
