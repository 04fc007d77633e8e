# dev: where does the host time of the shipped workload's forward go on a box?  (two of the round's profile runs read 13.9 ms per forward and a GPU 25 % idle, a direct run 0.3-0.5 ms)
R=$GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), {k: round(v,3) for k,v in d["timing"]["host_ms"].items()}, d["timing"]["gpu_idle_frac"], round(d["timing"]["host_ms_per_step_enqueue_loop"],2))'
uptime
cd $R && python3 bench.py --workload shipped --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "cwd=repo"
cd /tmp && python3 $R/bench.py --workload shipped --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "cwd=/tmp"
export TMPDIR=/tmp
cd /tmp && python3 $R/bench.py --workload shipped --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "cwd=/tmp,TMPDIR"
cd /tmp && python3 $R/bench.py --width 512 --steps 60 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "w512"
cd /tmp && python3 $R/bench.py --workload shipped --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "after w512"
uptime
