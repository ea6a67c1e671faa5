"""PLY / state_dict interchange (sk_gs_amd/io.py) -- CPU only.

The layout restates GaussianSplatting.save_ply / load_ply (networks/gaussian_splatting.py:340-428): attribute order,
channel-major SH flattening, raw parameter values.  The reference writes through the `plyfile` package, which is not
installed in the build image, so the byte stream is checked against a hand-assembled PLY instead of a reference file."""
import struct

import numpy as np
import torch

from sk_gs_amd import io, scene


def _params(P=7, seed=0):
    g = scene.make_gaussians(P, seed=seed, sh_degree=3)
    return {'_xyz': g['xyz'], '_features_dc': g['sh'][:, :1].contiguous(), '_features_rest': g['sh'][:, 1:].contiguous(),
            '_scaling': g['log_scale'], '_rotation': g['rot'], '_opacity': g['opacity_logit']}


def test_ply_round_trip_and_header(tmp_path):
    p = _params()
    path = str(tmp_path / 'sub' / 'pc.ply')
    io.save_ply(path, p)
    raw = open(path, 'rb').read()
    head, body = raw.split(b'end_header\n', 1)
    lines = head.decode().split('\n')
    assert lines[:3] == ['ply', 'format binary_little_endian 1.0', 'element vertex 7']
    names = [l.split()[-1] for l in lines if l.startswith('property float')]
    assert names == io.ply_attribute_names(3, 45) and len(names) == 62  # the reference's 62-float vertex
    assert len(body) == 7 * 62 * 4
    row0 = struct.unpack('<62f', body[:62 * 4])
    assert row0[:3] == tuple(p['_xyz'][0].tolist()) and row0[3:6] == (0.0, 0.0, 0.0)
    # SH are channel-major: f_rest_0..14 = coefficients 1..15 of channel R
    assert np.allclose(row0[9:9 + 15], p['_features_rest'][0, :, 0].numpy())
    assert np.allclose(row0[6:9], p['_features_dc'][0, 0].numpy())
    back = io.load_ply(path)
    for k in io.GAUSSIAN_PARAM_NAMES:
        assert back[k].shape == p[k].shape and torch.equal(back[k], p[k].float()), k


def test_reads_ascii_and_big_endian_ply(tmp_path):
    names = io.ply_attribute_names(3, 0)  # SH degree 0
    vals = np.arange(2 * len(names), dtype=np.float32).reshape(2, len(names)) * 0.5
    header = ['ply', 'format ascii 1.0', 'comment made by hand', 'element vertex 2'] + [f'property float {n}' for n in names]
    header += ['element face 0', 'property list uchar int vertex_indices', 'end_header']
    a = tmp_path / 'a.ply'
    a.write_text('\n'.join(header) + '\n' + '\n'.join(' '.join(repr(float(x)) for x in r) for r in vals) + '\n')
    header[1] = 'format binary_big_endian 1.0'
    b = tmp_path / 'b.ply'
    b.write_bytes(('\n'.join(header) + '\n').encode() + vals.astype('>f4').tobytes())
    for path in (a, b):
        got = io.load_ply(str(path), max_sh_degree=0)
        assert got['_features_rest'].shape == (2, 0, 3) and got['_features_dc'].shape == (2, 1, 3)
        assert torch.equal(got['_xyz'], torch.tensor(vals[:, :3]))
        assert torch.equal(got['_rotation'], torch.tensor(vals[:, -4:]))
        assert torch.equal(got['_opacity'][:, 0], torch.tensor(vals[:, 9]))


def test_state_dict_loads_by_reference_names():
    from sk_gs_amd.model import SkinnedGaussians
    m = SkinnedGaussians(50, 4, 2, num_frames=2)
    state = io.gaussians_state_dict(SkinnedGaussians(30, 4, 2, num_frames=2, seed=3))
    assert set(state) == set(io.GAUSSIAN_PARAM_NAMES)
    io.load_gaussians_state_dict(m, state)
    assert m.P == 30 and m._features_rest.shape == (30, 15, 3) and isinstance(m._xyz, torch.nn.Parameter)
    assert torch.equal(m._rotation.data, state['_rotation'])
