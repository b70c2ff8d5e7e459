cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('nt', b['ms_per_step'])"
MMSSL_LIB=$GRAFT_REPO_ROOT/tools/_dbg/libmmssl_plain.so timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('plain', b['ms_per_step'])"
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('driver cmd nt', b['ms_per_step'], b['projection']['forward'], b['projection']['weight_gradient'])"
