mkdir -p gpurun_out/r3a
python - > gpurun_out/r3a/host.txt 2>&1 <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:20])
for p in ("/sys/fs/cgroup/cpu.max","/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, e)
print("nproc", os.cpu_count())
try: print(open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read())
except Exception as e: print(e)
from onepose_st_amd import hostsize
print("rank_cpus(0,1)", hostsize.rank_cpus(0,1)); print("rank_cpus(3,8)", hostsize.rank_cpus(3,8))
PY
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a/tests.log
tail -5 gpurun_out/r3a/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench20.json 2> gpurun_out/r3a/bench20.err && tail -c 600 gpurun_out/r3a/bench20.json
timeout -k 10 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3a/bench200.json 2> gpurun_out/r3a/bench200.err
timeout -k 10 120 python tools/stamps_x3.py > gpurun_out/r3a/stamps_enc_x3w8.txt 2>&1
timeout -k 10 120 python tools/stamps_fine.py > gpurun_out/r3a/stamps_fine.txt 2>&1
timeout -k 10 120 python tools/time_fine.py > gpurun_out/r3a/time_fine.txt 2>&1
timeout -k 10 120 python tools/time_coarse.py > gpurun_out/r3a/time_coarse.txt 2>&1
echo done
