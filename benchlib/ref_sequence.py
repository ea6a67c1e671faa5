"""The reference's deform CALL SEQUENCES, restated for the tests that cannot import the reference (the GPU box has no
/root/reference): what ``SkeletonGaussianSplatting.forward(stage='sk' | 'sp')`` does between the parameters and the tensors handed
to the rasterizer, written against a ``lietorch``-shaped module and a ``knn_points`` -- in the tests: ``sk_gs_amd.lietorch`` and
``sk_gs_amd.pytorch3d_ops.knn_points``, i.e. exactly what the unmodified reference calls once ``install_as_lietorch()`` /
``install_as_pytorch3d()`` have run.  tests/golden/sk_stage.npz was produced by the reference's OWN code on those stand-ins
(tests/golden/make_golden_sk_stage.py); replaying it through these functions checks (a) that the restatement is the reference's
sequence and (b), on the GPU, that the stand-ins' HIP paths give what their CPU bodies gave.

Reference lines restated (networks/sk_gs.py): forward :1160-1204, sk_stage :1109-1150, kinematic :1069-1107, skeleton_warp_SE3
:193-206, calc_LBS_weight :751-774, sp_stage :830-856, warp :776-828.  Test infrastructure only.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _lbs_weights(knn_points, a, points, bones, K, feature=None, bone_feature=None):
    """calc_LBS_weight: the search, then one of the four weightings chosen by which parameters exist"""
    if feature is not None and bone_feature is not None:
        points = torch.cat([points.detach(), feature], dim=-1)
        bones = torch.cat([bones.detach(), bone_feature], dim=-1)
    nn_dist, indices, _ = knn_points(points[None], bones[None], None, None, K=K)
    nn_dist, indices = nn_dist[0], indices[0]
    if '_sp_radius' in a:
        radius = torch.exp(a['_sp_radius'])[indices]
        w = torch.exp(-nn_dist / (2 * radius ** 2))
        if '_sp_weight' in a:
            w = w * torch.sigmoid(a['_sp_weight'])[indices]
        w = w + 1e-7
        w = w / w.sum(dim=-1, keepdim=True)
    elif 'sp_W' in a:
        w = torch.gather(a['sp_W'], dim=1, index=indices).softmax(dim=-1)
    else:
        w = torch.softmax(-nn_dist / 1.0, dim=-1)
    return w, indices


def _activate(a, d_xyz, d_rot, d_scale):
    """the tail of forward(): what the rasterizer receives"""
    return {'points': a['_xyz'] + d_xyz, 'scales': torch.exp(a['_scaling']) + d_scale, 'rotations': F.normalize(a['_rotation'] + d_rot),
            'opacity': torch.sigmoid(a['_opacity'])}


def sk_stage(lie, knn_points, a, K):
    """stage `sk`; ``a``: dict of tensors named as in sk_stage.npz's ``<scenario>/in/`` (parameters may require grad)"""
    SE3, SO3 = lie.SE3, lie.SO3
    points = a['_xyz'].detach()
    g_tr = a['global_tr'][int(a['time_id'])].view(-1)
    joints, table, root = a['joints'], a['parents_table'], a['root']
    # kinematic: joint rotations -> local transforms about the joints -> chain
    sk_r, sk_d_rot, sk_d_scale = a['net_sk_r'], a['net_d_rot'], a['net_d_scale']
    if sk_r.shape[-1] == 4:
        sk_r = SO3.InitFromVec(F.normalize(sk_r + sk_r.new_tensor([0., 0., 0., 1.]), dim=-1))
    else:
        sk_r = SO3.exp(sk_r)
    sk_t = joints + sk_r.act(-joints)
    local = SE3.InitFromVec(torch.cat([sk_t, sk_r.vec()], dim=-1))
    out = local.vec().clone()
    out[root] = out.new_tensor([0, 0, 0, 0, 0, 0, 1.])
    out = SE3.InitFromVec(out)
    for level in range(table.shape[1]):
        out = out[table[:, level]] * out
    g = SE3.InitFromVec(g_tr)
    sk_T = g[None] * out
    # linear blend skinning
    w, idx = _lbs_weights(knn_points, a, points, joints, K)
    d_xyz = (sk_T[idx].act(points[:, None]) * w[..., None]).sum(dim=1) - points
    d_rot = (sk_d_rot[idx] * w[..., None]).sum(dim=1)
    d_scale = (sk_d_scale[idx] * w[..., None]).sum(dim=1)
    res = _activate(a, d_xyz, d_rot, d_scale)
    res.update(_skT=sk_T.vec(), _knn_w=w, _knn_i=idx, _sk_rot=sk_d_rot, _sk_scale=sk_d_scale, _d_xyz=d_xyz, _d_rot=d_rot, _d_scale=d_scale)
    return res


def sp_stage(lie, knn_points, a, K, warp_method='LBS', sep_rot=False):
    """stage `sp` (superpoints) with the three warp methods and the separate-rotation head"""
    SE3, SO3 = lie.SE3, lie.SO3
    points = a['_xyz'].detach()
    sp_points = a['sp_points']
    bias = points.new_tensor([0, 0, 0, 1.])
    w, idx = _lbs_weights(knn_points, a, points, sp_points, K, a['hyper_feature'], a['sp_hyper_feature'])
    d_xyz, d_scale = a['net_d_xyz'], a['net_d_scaling']
    d_rot = F.normalize(a['net_d_rotation'] + bias, dim=-1)
    g_rot = F.normalize(a['net_g_rotation'] + bias, dim=-1) if sep_rot else None
    p2sp = torch.gather(idx, -1, w.argmax(dim=-1, keepdim=True))[:, 0] if warp_method == 'largest' else None
    # warp
    sp_t = d_xyz
    if warp_method == 'LBS_c':
        sp_t = sp_t + sp_points + SO3.InitFromVec(d_rot).act(-sp_points)
    spT = SE3.InitFromVec(torch.cat([sp_t, d_rot], dim=-1))
    if warp_method in ('LBS', 'LBS_c'):
        d_points = (spT[idx].act(points[:, None]) * w[..., None]).sum(dim=1) - points
    else:
        d_points = spT[p2sp].act(points) - points
    blend_rot = g_rot if g_rot is not None else d_rot
    d_rotation = (blend_rot[idx] * w[..., None]).sum(dim=1)
    d_scales = (d_scale[idx] * w[..., None]).sum(dim=1)
    res = _activate(a, d_points, d_rotation, d_scales)
    res.update(_spT=spT.vec(), _knn_w=w, _knn_i=idx, _sp_scale=d_scale)
    if sep_rot:
        res['_sp_rot'] = g_rot
    if p2sp is not None:
        res['p2sp'] = p2sp
    return res


def loss_sp_arap(lie, spT, sp_points, sk_knn_num):
    """the reference's Lie-group regulariser on the superpoint transforms (loss_sp_arap, sk_gs.py:1371-1381): inverse, product, log
    and action of SE3 elements -- (mean geodesic distance to the neighbours' transforms, change of neighbour distances)"""
    se3 = lie.SE3.InitFromVec(spT)
    c = sp_points[..., :3]
    moved = se3.act(c)
    with torch.no_grad():
        dist = torch.cdist(c, c)
        _, knn = torch.topk(dist, dim=1, k=min(c.shape[0], sk_knn_num + 1), largest=False)
        knn = knn[:, 1:]
    loss = (se3[:, None].inv() * se3[knn]).log().norm(dim=-1).mean()
    dist_c = (c[:, None] - c[knn]).square().sum(dim=-1)
    dist_t = (moved[:, None] - moved[knn]).square().sum(dim=-1)
    return loss, (dist_c - dist_t).abs().mean()


# ---- stand-ins with the STRUCTURE of the reference's network classes (attribute and parameter names of networks/sk_gs.py:134-164,
# my_ext/blocks/mlp.py:43-85, networks/encoders/freq_encoder.py:61-75; tests/test_host_cpu.py runs the same accelerators' shadow
# builders on the reference's real classes in the build container)
class RefFreqEncoder(nn.Module):
    def __init__(self, input_dim, degree):
        super().__init__()
        self.input_dim, self.degree, self.output_dim = input_dim, degree, input_dim * (1 + 2 * degree)

    def forward(self, x):
        from sk_gs_amd.deform_net import freq_encode_torch
        return freq_encode_torch(x, self.degree)


class RefMLPWithSkips(nn.Module):
    def __init__(self, in_channels, dim_hidden, out_channels, num_layers, skips):
        super().__init__()
        self.in_channels, self.out_channels, self.dim_hidden, self.num_layers = in_channels, out_channels, dim_hidden, num_layers
        self.skips, self.bias, self.weight_norm = tuple(skips), True, False
        net, c = [], in_channels
        for i in range(num_layers):
            net.append(nn.Linear(c, dim_hidden))
            c = dim_hidden + (in_channels if i in self.skips else 0)
        self.net = nn.ModuleList(net)
        self.last = nn.ModuleList(nn.Linear(c, oc) for oc in out_channels)

    def forward(self, inputs):
        x = inputs
        for i in range(self.num_layers):
            x = F.relu(self.net[i](x))
            if i in self.skips:
                x = torch.cat([x, inputs], dim=-1)
        return [m(x) for m in self.last]


class RefSimpleDeformationNetwork(nn.Module):
    def __init__(self, out_channels=(4, 4, 3), p_degree=10, t_degree=6):
        super().__init__()
        self.pos_enc_p, self.pos_enc_t = RefFreqEncoder(3, p_degree), RefFreqEncoder(1, t_degree)
        self.dynamic_net = RefMLPWithSkips(self.pos_enc_p.output_dim + self.pos_enc_t.output_dim, 256, list(out_channels), 8, (4,))

    def forward(self, points, t):
        p_embed = self.pos_enc_p(points)
        t_embed = self.pos_enc_t(t.view(-1, 1)).expand(*points.shape[:-1], -1)
        return self.dynamic_net(torch.cat([p_embed, t_embed], dim=-1))


SCENARIOS = {  # name -> (stage, K, warp_method, sep_rot): tests/golden/make_golden_sk_stage.py
    'sk_W': ('sk', 5, 'LBS', False), 'sk_lie': ('sk', 5, 'LBS', False), 'sk_kernel': ('sk', 3, 'LBS', False),
    'sp_W_LBS': ('sp', 5, 'LBS', False), 'sp_wk_LBSc_sep': ('sp', 3, 'LBS_c', True), 'sp_W_largest': ('sp', 3, 'largest', False),
    'sp_dist_LBS': ('sp', 5, 'LBS', False), 'sp_kernel_LBSc': ('sp', 4, 'LBS_c', False),
}


def load_scenario(npz, name, device='cpu'):
    """(inputs with requires_grad set on every float tensor that has a recorded gradient, expected outputs, cotangents, expected grads)"""
    pre = name + '/'
    a, out, cot, grad = {}, {}, {}, {}
    for key in npz.files:
        if not key.startswith(pre):
            continue
        _, kind, k = key.split('/')
        t = torch.from_numpy(npz[key])
        {'in': a, 'out': out, 'cot': cot, 'grad': grad}[kind][k] = t
    for k in list(a):
        t = a[k].to(device)
        if k in grad:
            t.requires_grad_()
        a[k] = t
    return a, out, cot, grad


def run_scenario(lie, knn_points, npz, name, device='cpu'):
    stage, K, warp, sep = SCENARIOS[name]
    a, out, cot, grad = load_scenario(npz, name, device)
    res = sk_stage(lie, knn_points, a, K) if stage == 'sk' else sp_stage(lie, knn_points, a, K, warp, sep)
    loss = 0
    for k, G in cot.items():
        if res[k].requires_grad:
            loss = loss + (res[k] * G.to(device)).sum()
    loss.backward()
    got_grad = {k: a[k].grad for k in grad}
    return res, out, got_grad, grad
