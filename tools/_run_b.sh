cd $GRAFT_REPO_ROOT
python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-c5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); fc=d['frame_chain']
print('default: ms_per_step %.4f value %.3g | frame %.4f dropin %.4f eager %.3f'%(d['ms_per_step'],d['value'],fc['ms_per_frame'],fc['dropin_ms_per_frame'],fc['dropin_eager_ms_per_frame']))"
VGICP_UPLOAD_STAGE_LIMIT=16777216 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-c5 --no-frame-chain 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('upload staged (16 MB limit): ms_per_step %.4f value %.3g'%(d['ms_per_step'],d['value']))"
