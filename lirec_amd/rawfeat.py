"""Raw-feature access on the device: ``.npy`` reading, spatial / person-box mean, temporal max.

The reference computes a clip or track feature that is not in its disk cache with numpy on the host
(visual_utils/visual_features.py, text_utils/text_features.py, mixed_utils/mixed_features.py:37-112): it loads the
scene's I3D grid ``[F, 2048, H, W]`` and BERT token matrix from ``.npy``, averages the grid over space (clip) or over a
person box derived from the face track (track), and takes the maximum over the clip's frames / the track's elements /
the tokens of the time range.  Here the grid and the token matrix are read straight into pinned memory and kept in HBM,
the small integer bookkeeping (frame ranges, box corners, token ranges) stays on the host exactly as the reference
writes it, and the reductions are two kernels (``lirec_grid_pool``, ``lirec_rows_max``) whose results equal numpy's bit
for bit (tests/test_rawfeat.py, fixture produced by the reference's own classes).
"""
from __future__ import annotations

import ast
import ctypes as C
import struct

import numpy as np
import torch

from ._lib import check, lib


# ---------------------------------------------------------------------------
# .npy
# ---------------------------------------------------------------------------

def npy_header(f):
    """Parse a ``.npy`` header (format 1.0 / 2.0 / 3.0): returns (dtype, fortran_order, shape, data offset)."""
    magic = f.read(6)
    if magic != b'\x93NUMPY':
        raise ValueError('not a .npy file')
    major, _minor = struct.unpack('BB', f.read(2))
    hlen = struct.unpack('<H', f.read(2))[0] if major == 1 else struct.unpack('<I', f.read(4))[0]
    d = ast.literal_eval(f.read(hlen).decode('latin1' if major < 3 else 'utf8'))
    return np.dtype(d['descr']), bool(d['fortran_order']), tuple(d['shape']), f.tell()


def load_npy(path, device='cuda', dtype=torch.float32):
    """``np.load(path)`` into HBM: the file's bytes are read once into a pinned buffer of the file's own dtype and
    copied to the device, then converted there when the dtype differs (the reference keeps these arrays as numpy on the
    host, visual_features.py:34-38, text_features.py:106-118)."""
    with open(path, 'rb') as f:
        dt, fortran, shape, off = npy_header(f)
        if fortran or dt.hasobject or dt.byteorder == '>':
            raise ValueError('unsupported .npy layout: %s fortran=%s' % (dt, fortran))
        n = int(np.prod(shape, dtype=np.int64))
        tdt = {np.dtype('<f4'): torch.float32, np.dtype('<f8'): torch.float64, np.dtype('<f2'): torch.float16,
               np.dtype('<i4'): torch.int32, np.dtype('<i8'): torch.int64}[dt]
        pin = torch.device(device).type == 'cuda'
        host = torch.empty(n, dtype=tdt, pin_memory=pin)
        f.seek(off)
        got = f.readinto(memoryview(host.numpy()).cast('B'))
        if got != n * dt.itemsize:
            raise ValueError('%s: truncated (%d of %d bytes)' % (path, got, n * dt.itemsize))
    t = host.view(shape).to(device, non_blocking=True)
    return t if dtype is None or t.dtype == dtype else t.to(dtype)


# ---------------------------------------------------------------------------
# host bookkeeping, as the reference writes it
# ---------------------------------------------------------------------------

def clip_frame_range(time2frame: dict, time_node: dict, n_frames: int, sampling_fr):
    """Frames of a clip (visual_features.py:75-95): first frame of the start second .. last frame of the end second,
    scaled by ``sampling_fr`` when it is below 1 (the shipped recipes: 1/16), clipped to the grid's length."""
    start = time2frame[int(time_node['start'])][0]
    end_time = int(time_node['end'])
    end_time = end_time if end_time in time2frame else end_time - 1          # "due to problems with time rounding"
    end = time2frame[end_time][-1]
    if sampling_fr < 1:
        start, end = int(start * sampling_fr), int(end * sampling_fr)
    step = 1 if sampling_fr < 1 else sampling_fr
    if end >= n_frames:
        return list(range(start, n_frames, step))
    return list(range(start, end + 1, step))


def track_boxes(track, dims, H, W, n_frames, sampling_fr):
    """[frame, y0, y1, x0, x1] per track element (visual_features.py:110-131): the face box is blown up to a person box
    (face = 0.35..0.65 of its width, 0.10..0.25 of its height), scaled to the I3D grid, floor / ceil to cells.  A frame
    index equal to the grid length marks the element the reference skips (its row stays zero): frame -1 here."""
    sh, sw = H / dims[0], W / dims[1]
    FH0, FH1, FW0, FW1 = 0.10, 0.25, 0.35, 0.65
    out = []
    for el in track:
        fx, fy, fw, fh = el['x'] / 2., el['y'] / 2., el['w'] / 2., el['h'] / 2.
        pw, ph = fw / (FW1 - FW0), fh / (FH1 - FH0)
        px, py = fx - FW0 * pw, fy - FH0 * ph
        spx, spw, spy, sph = px * sw, pw * sw, py * sh, ph * sh
        rx = [max(0, int(np.floor(spx))), min(int(W), int(np.ceil(spx + spw)))]
        ry = [max(0, int(np.floor(spy))), min(int(H), int(np.ceil(spy + sph)))]
        frame = int(el['frame'] * sampling_fr)
        out.append([-1 if frame == n_frames else frame, ry[0], ry[1], rx[0], rx[1]])
    return out


def token_rows(times, time_idx2token_range, time_node):
    """Token rows of the dialog lines that overlap a clip (text_features.py:150-160); ``times``: (start, end) seconds."""
    rows = []
    s, e = time_node['start'], time_node['end']
    for i, (ts, te) in enumerate(times):
        if ts <= s <= te or ts <= e <= te or (s <= ts and e >= te):        # Time.includes, :24-31
            rows += list(time_idx2token_range[i])
    return rows


# ---------------------------------------------------------------------------
# device reductions
# ---------------------------------------------------------------------------

def _stream():
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _segments(lists, width, device):
    flat = [x for l in lists for x in l]
    est = np.zeros(len(lists) + 1, dtype=np.int32)
    est[1:] = np.cumsum([len(l) for l in lists])
    arr = torch.tensor(flat, dtype=torch.int32).reshape(-1, width) if flat else torch.zeros((0, width), dtype=torch.int32)
    return arr.to(device).contiguous(), torch.from_numpy(est).to(device)


def grid_pool(grid: torch.Tensor, box_lists) -> torch.Tensor:
    """``grid`` [F, C, H, W] fp32 on the device; ``box_lists``: per output a list of [frame, y0, y1, x0, x1].
    Returns [len(box_lists), C]: max over the list of the box means."""
    assert grid.is_cuda and grid.dtype == torch.float32 and grid.dim() == 4 and grid.is_contiguous()
    F, Cc, H, W = grid.shape
    boxes, est = _segments(box_lists, 5, grid.device)
    out = torch.empty((len(box_lists), Cc), dtype=torch.float32, device=grid.device)
    check(lib().lirec_grid_pool(grid.data_ptr(), F, Cc, H, W, boxes.data_ptr() if boxes.numel() else est.data_ptr(), est.data_ptr(),
                                len(box_lists), out.data_ptr(), Cc, _stream()), 'lirec_grid_pool')
    return out


def rows_max(src: torch.Tensor, row_lists) -> torch.Tensor:
    """``src`` [N, dim] fp32 (may be a strided view with unit column stride); returns [len(row_lists), dim]."""
    assert src.is_cuda and src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    idx, est = _segments([[[r] for r in l] for l in row_lists], 1, src.device)
    out = torch.empty((len(row_lists), src.shape[1]), dtype=torch.float32, device=src.device)
    check(lib().lirec_rows_max(src.data_ptr(), src.stride(0), idx.data_ptr() if idx.numel() else est.data_ptr(), est.data_ptr(),
                               len(row_lists), src.shape[1], out.data_ptr(), src.shape[1], _stream()), 'lirec_rows_max')
    return out


def clip_visual_features(grid, time2frame, time_nodes, sampling_fr=0.0625):
    """[len(time_nodes), C]: f_visual(visual.get_features_by_time(t)) of mixed_features.py:54 for every clip."""
    F, _, H, W = grid.shape
    return grid_pool(grid, [[[f, 0, H, 0, W] for f in clip_frame_range(time2frame, t, F, sampling_fr)] for t in time_nodes])


def track_features(grid, tracks, dims, sampling_fr=0.0625):
    """[len(tracks), C]: f_visual(visual.get_features_by_track(track)) of mixed_features.py:104-105 (opt.tf_crop)."""
    F, _, H, W = grid.shape
    return grid_pool(grid, [track_boxes(tr, dims, H, W, F, sampling_fr) for tr in tracks])


def clip_text_features(tokens, times, time_idx2token_range, time_nodes, layer=-2):
    """[len(time_nodes), text_dim]: f_text(textual.get_features_by_time(t)) of mixed_features.py:61 on the
    ``[tokens, layers, text_dim]`` BERT dump with the 'second-to-last' contextualisation (text_features.py:181-182;
    ``layer=-1``: 'last').  A clip without dialog gets zeros (:172-177)."""
    assert tokens.dim() == 3
    view = tokens[:, layer, :]
    return rows_max(view, [token_rows(times, time_idx2token_range, t) for t in time_nodes])
