import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd import _C, scene
from sk_gs_amd.losses import image_loss
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.train_step import GraphedSteps
from sk_gs_amd.view_parallel import ViewParallel
dev = torch.device('cuda')
P, M, K, W, H = [int(x) for x in os.environ.get("DBG_SHAPE", "20000,20,5,256,256").split(",")]
model = SkinnedGaussians(P, M, K, num_frames=int(os.environ.get("DBG_FRAMES", "2")), seed=0, scale_mult=float(os.environ.get("DBG_SCALE", "2.0"))).to(dev)
cam = scene.make_camera(W, H, seed=0)
rs = scene.raster_settings_from_camera(cam, colmap=True, device=dev)
bg = torch.ones(3, device=dev)
with torch.no_grad():
    target = model.render(rs, 0, bg)['images'].clone()
R = model.render(rs, 0, bg)['buffer'].R
print('R', R, flush=True)
_C.config.sync_num_rendered = False
_C.update_capacity_hint(P, W, H, R * 2)
stage = sys.argv[1] if len(sys.argv) > 1 else 'all'
opt = torch.optim.Adam(model.param_groups(lr=1e-4), eps=1e-15, fused=True, capturable=True)
vp = ViewParallel(model.parameters())

def fwd_only(v):
    with torch.no_grad():
        out = model.render(rs, 0, bg)

def fwd_bwd(v):
    vp.grads.zero_()
    out = model.render(rs, 0, bg)
    loss = image_loss(out['images'], target)
    loss.backward()

def full(v):
    fwd_bwd(v)
    opt.step()

fn = dict(fwd=fwd_only, fwdbwd=fwd_bwd, all=full)[stage]
full(0)
torch.cuda.synchronize()
print('eager ok', flush=True)
g = GraphedSteps(fn)
g(0)
torch.cuda.synchronize()
print('capture+replay ok', flush=True)
for _ in range(5):
    g(0)
torch.cuda.synchronize()
print('done', stage, flush=True)

# ---- multi-view variants
mode = sys.argv[2] if len(sys.argv) > 2 else None
if mode:
    cams = [scene.make_camera(W, H, seed=i) for i in range(3)]
    rss = [scene.raster_settings_from_camera(c, colmap=True, device=dev) for c in cams]
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    def mv(v):
        vp.grads.zero_()
        out = model.render(rss[v], v % 2, bg)
        loss = image_loss(out['images'], target)
        loss.backward()
        overflow.add_(out['buffer'].geomBuffer[4:8].view(torch.int32))
        opt.step()
    gs = GraphedSteps(mv)
    if mode == 'nopool':
        class NP(GraphedSteps):
            def capture(self, key):
                self.pool = None
                return super().capture(key)
        gs = NP(mv)
    for i in range(12):
        gs(i % 3)
    torch.cuda.synchronize()
    print('multi ok', mode, int(overflow.item()), flush=True)
