set -e
mkdir -p gpurun_out
TLN_GRU_ROWS64=1 timeout -k 10 600 python -m pytest tests/test_gpu_gemm_v2.py tests/test_gpu_streams.py tests/test_gpu_engine.py tests/test_gpu_ops.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r64_tests.log 2>&1 || { tail -30 gpurun_out/r64_tests.log; exit 1; }
tail -2 gpurun_out/r64_tests.log
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --frames-extra off > gpurun_out/ab_$NAME.json 2>gpurun_out/ab_$NAME.err; }
run2() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --frames-extra off --streams 1 --pairs 0 > gpurun_out/ab_$NAME.json 2>gpurun_out/ab_$NAME.err; }
NAME=new1 run A=1 && NAME=gru1 run TLN_GRU_ROWS64=1 && NAME=new2 run A=1 && NAME=gru2 run TLN_GRU_ROWS64=1 && NAME=snew run2 A=1 && NAME=sgru run2 TLN_GRU_ROWS64=1
python - <<'PY'
import json
for n in ("new1","gru1","new2","gru2","snew","sgru"):
    d=json.loads(open("gpurun_out/ab_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["roofline"]["frac"], d["roofline"]["executed"]["frac"], d["roofline"]["one_sequence_alone"]["frac"])
PY
