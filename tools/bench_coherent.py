"""Headline workload with the Gaussians in the benchmark's uniformly random order vs in row-major pixel order, which is how
scripts/hierslam.py creates them from an RGB-D frame (hierslam.py:361-389): spatially coherent input lets the scattered stores
of the tile binning and the record gathers of the tile kernels coalesce.  Prints renders/s for both."""
import sys, time, numpy as np, torch
sys.path.insert(0,'hier-slam_amd')
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer_semantic, _C
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
from hsr_utils.synthetic import make_scene, make_upstream_grads
W,H,K,P=1200,680,26,500000
k=replica_intrinsics(W,H); cam_cpu=setup_camera_tensors(W,H,k,np.eye(4)); dev=torch.device('cuda')
cam=GaussianRasterizationSettings(**{kk:(v.to(dev) if isinstance(v,torch.Tensor) else v) for kk,v in cam_cpu.items()})
sc=make_scene(P,W,H,K,k,seed=0)
up=make_upstream_grads(W,H,K,seed=1); upd=[up[n].to(dev) for n in ("color","semantic","depth","median","opacity")]
names=("means3D","colors_precomp","semantics_precomp","opacities","scales","rotations")
for order in ("random","pixel"):
    if order=="pixel":
        m=sc["means3D"]; u=(m[:,0]/m[:,2]*600+599.5); v=(m[:,1]/m[:,2]*600+339.5)
        key=(v.floor().clamp(-16,H+16)+16)*(W+64)+u   # row-major pixel order, like an unprojected RGB-D frame
        idx=torch.argsort(key)
        s2={n:sc[n][idx].contiguous() for n in names}
    else: s2=sc
    leaf={n:s2[n].to(dev).requires_grad_(True) for n in names}
    r=GaussianRasterizer_semantic(cam)
    def step():
        m2=torch.zeros(P,3,device=dev,requires_grad=True)
        outs=r(means3D=leaf["means3D"],means2D=m2,opacities=leaf["opacities"],colors_precomp=leaf["colors_precomp"],scales=leaf["scales"],rotations=leaf["rotations"],semantics_precomp=leaf["semantics_precomp"])
        for n in names: leaf[n].grad=None
        torch.autograd.backward([outs[0],outs[2],outs[3],outs[4],outs[5]],upd)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
    print(order, "renders/s %.1f  ms %.4f" % (1/dt, dt*1e3))
