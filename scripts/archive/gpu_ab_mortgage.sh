cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in nvcomp_amd/lib/alt/libnvcomp_*.so; do
  tag=$(basename $lib .so)
  NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --dataset mortgage_col0_like --producer fast --unique-mib 314 --mib-per-gpu 314 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$tag mortgage', r['value'])"
done
