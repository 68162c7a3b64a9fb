#!/bin/bash
# round 5: the N>1 bench line at FULL size with two real processes on the one GPU (the stream-ordered librccl stand-in of the tests): not a scaling number — both ranks
# share one GPU — but the whole N>1 control flow of the default command at the headline size: calibration, the timed region, latency, exchange_forms, the watchdog
gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/tools/fake_rccl.c -o /tmp/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
for ex in gather host; do
  LPT_RCCL_LIBRARY=/tmp/libfake_rccl.so timeout 900 python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 --exchange $ex > gpurun_out/r05y_2proc_$ex.json 2> gpurun_out/r05y_2proc_$ex.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r05y_2proc_$ex.json").read().strip().splitlines()[-1])
    ef = j["exchange_forms"]
    print("$ex: n_gpus", j["n_gpus"], "ms/frame %.3f" % j["ms_per_frame"], "frame_complete", j["config"]["frame_complete"], "forms", {k: (round(v["ms_per_frame"], 3) if "ms_per_frame" in v else v) for k, v in ef.items() if isinstance(v, dict)}, "checksums_equal", ef.get("checksums_equal"))
    print("   stage_ms_per_rank", {k: (round(v["min"], 3), round(v["max"], 3)) for k, v in j["stage_ms_per_rank"].items()})
    print("   rccl", {k: j["rccl"][k] for k in ("rccl_nranks", "tile_weights", "exchange_ms_per_frame_rank0")} if j["rccl"] else None)
except Exception as e:
    print("$ex: FAILED", e); print(open("gpurun_out/r05y_2proc_$ex.err").read()[-1500:])
PY
done
