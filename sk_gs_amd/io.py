"""Interchange formats of the Gaussian parameters (scope row (f)-4: data formats on either side of the path).

``save_ply`` / ``load_ply`` restate ``GaussianSplatting.save_ply`` / ``load_ply`` (networks/gaussian_splatting.py:340-428):
one ``vertex`` element with float32 properties

    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3*(S-1)-1)  opacity  scale_0..2  rot_0..3

in the *channel-major* SH order of the reference (``features.transpose(1, 2).flatten(1)``: all coefficients of R, then G,
then B), raw (pre-activation) opacity / scale / rotation values, zero normals.  The reference writes through the
``plyfile`` package (``PlyData([el]).write(path)`` -> ``format binary_little_endian 1.0``); that package is not a
dependency here, so the 1.0 binary layout is written / parsed directly (ASCII files are read too).  Parity of the byte
stream against a ``plyfile``-written file is unpinned (the package is absent from the build image); the round trip and the
header are tested.

``gaussians_state_dict`` / ``load_gaussians_state_dict`` use the reference's parameter names (``_xyz``, ``_features_dc``,
``_features_rest``, ``_scaling``, ``_rotation``, ``_opacity``; gaussian_splatting.py:134-139,430-441), so a checkpoint
``state_dict`` of the reference's Gaussian module loads by key.
"""
import os
from typing import Dict, Mapping

import numpy as np
import torch
from torch import Tensor, nn

GAUSSIAN_PARAM_NAMES = ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity')


def ply_attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4):
    """``construct_list_of_attributes`` (gaussian_splatting.py:340-353)"""
    names = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    names += [f'f_dc_{i}' for i in range(n_dc)]
    names += [f'f_rest_{i}' for i in range(n_rest)]
    names.append('opacity')
    names += [f'scale_{i}' for i in range(n_scale)]
    names += [f'rot_{i}' for i in range(n_rot)]
    return names


def save_ply(path: str, params: Mapping[str, Tensor]):
    """``params``: the six raw parameter tensors by the reference's names ([P,3], [P,1,3], [P,S-1,3], [P,3], [P,4], [P,1])"""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    g = {k: params[k].detach().float().cpu() for k in GAUSSIAN_PARAM_NAMES}
    xyz = g['_xyz'].numpy()
    f_dc = g['_features_dc'].transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    f_rest = g['_features_rest'].transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, g['_opacity'].numpy().reshape(len(xyz), -1),
                           g['_scaling'].numpy(), g['_rotation'].numpy()), axis=1).astype('<f4')
    names = ply_attribute_names(f_dc.shape[1], f_rest.shape[1], g['_scaling'].shape[1], g['_rotation'].shape[1])
    assert cols.shape[1] == len(names)
    header = ['ply', 'format binary_little_endian 1.0', f'element vertex {cols.shape[0]}']
    header += [f'property float {n}' for n in names]
    header.append('end_header')
    with open(path, 'wb') as f:
        f.write(('\n'.join(header) + '\n').encode('ascii'))
        f.write(np.ascontiguousarray(cols).tobytes())


_PLY_TYPES = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1',
              'char': 'i1', 'int8': 'i1', 'short': '<i2', 'int16': '<i2', 'ushort': '<u2', 'uint16': '<u2',
              'int': '<i4', 'int32': '<i4', 'uint': '<u4', 'uint32': '<u4'}


def read_ply_vertices(path: str) -> Dict[str, np.ndarray]:
    """property name -> column of the first ``vertex`` element (binary little/big endian or ASCII PLY 1.0)"""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'{path}: not a PLY file')
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f'{path}: unterminated PLY header')
            tok = line.decode('ascii').split()
            if not tok or tok[0] == 'comment':
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex' and count is None
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError('list properties on the vertex element are not supported')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        if fmt is None or count is None:
            raise ValueError(f'{path}: missing format / vertex element')
        if fmt == 'ascii':
            rows = np.loadtxt(f, max_rows=count, dtype=np.float64, ndmin=2)
            return {n: rows[:, i].astype(np.dtype(t).newbyteorder('=')) for i, (n, t) in enumerate(props)}
        swap = fmt == 'binary_big_endian'
        dt = np.dtype([(n, np.dtype(t).newbyteorder('>') if swap else t) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        return {n: np.ascontiguousarray(data[n]) for n, _ in props}


def load_ply(path: str, max_sh_degree: int = 3, device='cpu') -> Dict[str, Tensor]:
    """The six raw parameter tensors of a PLY written by the reference (or by ``save_ply``); shapes as in
    ``GaussianSplatting.load_ply`` (gaussian_splatting.py:383-428)."""
    v = read_ply_vertices(path)
    P = len(v['x'])
    xyz = np.stack((v['x'], v['y'], v['z']), axis=1)
    features_dc = np.stack((v['f_dc_0'], v['f_dc_1'], v['f_dc_2']), axis=1)[..., None]          # [P,3,1]
    rest_names = sorted((n for n in v if n.startswith('f_rest_')), key=lambda n: int(n.split('_')[-1]))
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f'{path}: {len(rest_names)} f_rest properties do not match SH degree {max_sh_degree}')
    n_rest = (max_sh_degree + 1) ** 2 - 1
    features_rest = (np.stack([v[n] for n in rest_names], axis=1) if rest_names else np.zeros((P, 0))).reshape(P, 3, n_rest)
    scale_names = sorted((n for n in v if n.startswith('scale_')), key=lambda n: int(n.split('_')[-1]))
    rot_names = sorted((n for n in v if n.startswith('rot')), key=lambda n: int(n.split('_')[-1]))

    def t(a):
        return torch.tensor(np.asarray(a, dtype=np.float32), dtype=torch.float32, device=device)

    return {
        '_xyz': t(xyz),
        '_features_dc': t(features_dc).transpose(1, 2).contiguous(),
        '_features_rest': t(features_rest).transpose(1, 2).contiguous(),
        '_opacity': t(v['opacity'])[..., None],
        '_scaling': t(np.stack([v[n] for n in scale_names], axis=1)),
        '_rotation': t(np.stack([v[n] for n in rot_names], axis=1)),
    }


def gaussians_state_dict(module: nn.Module) -> Dict[str, Tensor]:
    return {k: getattr(module, k).detach().clone() for k in GAUSSIAN_PARAM_NAMES}


def load_gaussians_state_dict(module: nn.Module, state: Mapping[str, Tensor]):
    """Replace the six Gaussian parameters by the tensors of ``state`` (shapes may change: the reference re-creates the
    parameters to the checkpoint's shapes before loading, gaussian_splatting.py:430-441).  Any optimizer / gradient
    buffer bound to the old parameters must be rebuilt by the caller."""
    ref = getattr(module, '_xyz')
    P = state['_xyz'].shape[0]
    for k in GAUSSIAN_PARAM_NAMES:
        if state[k].shape[0] != P:
            raise ValueError(f'{k}: {state[k].shape[0]} rows, expected {P}')
        setattr(module, k, nn.Parameter(state[k].detach().to(device=ref.device, dtype=torch.float32).contiguous().clone()))
    if hasattr(module, 'P'):
        module.P = P
    if getattr(module, 'capacity', None) is not None:
        # the new parameters are plain tensors: the row capacity (sk_gs_amd/capacity.py) no longer describes them.  Re-home
        # with module.enable_capacity(...) before rebuilding gradient buffers / optimizer / step
        module.capacity = None
