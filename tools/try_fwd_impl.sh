# parity cases + headline bench with HSR_FWD_IMPL=$1 (usage: bash tools/try_fwd_impl.sh pair)
# (kernel-family / A-B selectors and ablation switches live in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
export HSR_FWD_IMPL=$1
python - <<'PY'
import sys; sys.path[:0]=['hier-slam_amd','tests']
import scenes
from test_gpu_parity import CASES, _compare
for n in ('replica_tree_k26','scannet_tree_k16','generic_k5_white_bg','plain_mask','plain_cov3d','huge_splats','deep_tiles_3000','semantic_k0','culled_behind_camera'):
    W,H,P,K,kind,sm,sem,var,bg,beh = CASES[n]
    cam,sc,up = scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)
    _compare(cam,sc,up,sem,var,None); print('ok', n)
PY
python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/try_fwd.json && python -c "
import json;d=json.load(open('gpurun_out/try_fwd.json'));print(round(d['value'],1), {a:round(b,3) for a,b in d['stages_ms'].items()})"
