# usage: ab.sh <libsuffix...>   -- same-box A/B by the bench's kernel time
for r in 1 2; do for v in "$@"; do
  [ "$v" = "cur" ] && L=/root/repo/frank_amd/libfrank_hip.so || L=/root/repo/frank_amd/libfrank_hip$v.so
  echo -n "$v: "; FRANK_AMD_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['breakdown_ms']['fit_loop_kernel'],2))"
done; done
