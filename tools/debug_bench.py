"""bisect helper: bench-like flow with toggles (argv letters)"""
import faulthandler, sys, os
if not os.environ.get("NO_FH"): faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
flags = sys.argv[1] if len(sys.argv) > 1 else ''
if 'd' in flags:
    import torch.distributed as dist
from sk_gs_amd import _C, scene
from sk_gs_amd.losses import image_loss
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.train_step import GraphedSteps
from sk_gs_amd.view_parallel import ViewParallel, init_distributed
if 'i' in flags:
    init_distributed()
if 's' in flags:
    torch.cuda.set_device(0)
dev = torch.device('cuda', 0) if 'x' in flags else torch.device('cuda')
if 'l' in flags:
    _C.load_library()
P, M, K, W, H = (20000, 20, 5, 256, 256) if 'm' in flags else (10000, 0, 0, 400, 400)
views = 1
model = SkinnedGaussians(P, M, K, sh_degree=3, num_frames=views, seed=0).to(dev)
cams = [scene.make_camera(W, H, seed=i) for i in range(views)]
settings = [scene.raster_settings_from_camera(c, sh_degree=3, colmap=True, device=dev) for c in cams]
background = torch.ones(3, device=dev)
gen = torch.Generator().manual_seed(77)
targets = []
with torch.no_grad():
    for v in range(views):
        img = model.render(settings[v], time_id=v % views, background=background)['images']
        targets.append(img.clone() if 'r' in flags else (img + 0.05 * torch.randn(3, H, W, generator=gen).to(dev)).clamp(0, 1).contiguous())
opt = torch.optim.Adam(model.param_groups(lr=1e-4), eps=1e-15, betas=(0.9, 0.999), fused=True, capturable=True)
vp = ViewParallel(model.parameters(), average=True)
overflow = torch.zeros(1, dtype=torch.int32, device=dev)

def fwd_bwd(v):
    vp.grads.zero_()
    out = model.render(settings[v], time_id=v % views, background=background)
    loss = image_loss(out['images'], targets[v])
    loss.backward()
    if 'a' not in flags:
        overflow.add_(out['buffer'].geomBuffer[4:8].view(torch.int32))

def eager_step(i):
    fwd_bwd(0)
    vp.allreduce_grads()
    opt.step()

if 'L' in flags:
    g_step = GraphedSteps(lambda v: (fwd_bwd(v), opt.step()))
else:
    def full(v):
        fwd_bwd(v)
        opt.step()
    g_step = GraphedSteps(full)
_C.config.sync_num_rendered = True
Rs = []
import contextlib
with (torch.no_grad() if 'n' in flags else contextlib.nullcontext()):
    for v in range(views):
        out = model.render(settings[v], time_id=v % views, background=background)
        Rs.append(out['buffer'].R)
if 'o' in flags:
    del out
_C.config.sync_num_rendered = False
_C.update_capacity_hint(P, W, H, int(max(Rs) * 1.25))
eager_step(0)
for i in range(4):
    g_step(0)
torch.cuda.synchronize()
print('OK', flags, flush=True)
