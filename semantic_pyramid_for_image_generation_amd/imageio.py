"""Host-side I/O the CLIs need and torchvision / tensorboardX used to provide (neither is installed
here): an ImageFolder with Resize(256)+CenterCrop(256)+ToTensor -- on the host through PIL, or
(raw=True + device_batch) decoded on the host and resized / cropped / converted / normalised by ONE GPU launch per image,
bit-identical to the PIL path --, a sharded loader, make_grid / save_image, and a scalar writer.  None of this is on the timed
path."""
import os
import time

import numpy as np
import torch

IMG_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


class ImageFolder(torch.utils.data.Dataset):
    """root/<class>/<image> -> (float tensor (3,crop,crop) in [0,1], class index); same transform chain as
    train_model.py:233-238 (shorter side -> image_size, bilinear with antialias; center crop; ToTensor)."""

    def __init__(self, root, image_size=256, crop_size=256, raw=False):
        # raw: __getitem__ returns the decoded uint8 (height, width, 3) pixels; resize / crop / ToTensor / Normalize then run on
        # the GPU (device_transform), batch them with raw_collate
        self.raw = raw
        classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(classes)}
        self.imgs = []
        for c in classes:
            for dp, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXT):
                        self.imgs.append((os.path.join(dp, f), self.class_to_idx[c]))
        self.image_size, self.crop_size = image_size, crop_size

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, index):
        from PIL import Image
        path, target = self.imgs[index]
        with open(path, 'rb') as f:
            img = Image.open(f).convert('RGB')
        if self.raw:
            return torch.from_numpy(np.array(img, dtype=np.uint8)), target
        w, h = img.size
        s = self.image_size
        if w <= h:
            nw, nh = s, int(s * h / w)
        else:
            nw, nh = int(s * w / h), s
        img = img.resize((nw, nh), Image.BILINEAR)
        left, top = int(round((nw - self.crop_size) / 2.0)), int(round((nh - self.crop_size) / 2.0))
        img = img.crop((left, top, left + self.crop_size, top + self.crop_size))
        arr = np.array(img, dtype=np.uint8)
        return torch.from_numpy(arr).permute(2, 0, 1).float().div(255), target


# ------------------------------------------------------------------------------------------------
# the same transform chain on the GPU (sempyr_resize_crop_normalize): decoded uint8 pixels in, float planes out
# ------------------------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2


def resample_table(in_size, out_size, first, count):
    """Integer coefficient table of PIL's bilinear ImagingResample (8-bit images) for output positions first .. first+count-1
    of an in_size -> out_size resize: int32 array [count][2 + ksize] = {first source index, taps, coefficients}.  Follows
    Pillow's precompute_coeffs / normalize_coeffs_8bpc step by step in double precision (support = max(scale, 1), window
    [int(center - support + 0.5), int(center + support + 0.5)), triangle weights normalised to 1, 22 fractional bits rounded
    to nearest), so the device kernel reproduces PIL's pixels exactly."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    tab = np.zeros((count, 2 + ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for i in range(count):
        xx = first + i
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        tab[i, 0], tab[i, 1] = xmin, xmax
        for x, v in enumerate(w):
            tab[i, 2 + x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
    return tab, ksize


_TABLES = {}
# the CLIs decode on the host and transform on the GPU unless SEMPYR_DEVICE_TRANSFORM=0 (then PIL does the whole chain)
DEVICE_TRANSFORM = os.environ.get("SEMPYR_DEVICE_TRANSFORM", "1") != "0"


def batch_images(data, device):
    """The (B,3,crop,crop) float batch in [0,1] on `device` from what a loader of ImageFolder yields: a tensor batch (host
    transform chain) is uploaded, a list of raw decoded images goes through device_batch."""
    if isinstance(data, (list, tuple)):
        return device_batch(data, device)[0]
    return data.to(device, non_blocking=True)


def device_transform(pixels, device, image_size=256, crop_size=256, mean_a=None, std_a=None, mean_b=None, std_b=None,
                     want01=True, out=None):
    """Resize(image_size) -> CenterCrop(crop_size) -> ToTensor [-> Normalize(mean_a, std_a)] [-> Normalize(mean_b, std_b)] of ONE
    decoded image (uint8 array / tensor, height x width x 3) on the GPU, one launch (train_model.py:233-247,308-309).
    Returns (image in [0,1] or None, copy a or None, copy b or None), each (3, crop, crop) float32 on `device`; `out`: a triple
    of existing tensors to write into (e.g. rows of a batch).  The pixels are bit-identical to the PIL path of ImageFolder."""
    import ctypes
    from . import _lib
    t = torch.as_tensor(pixels)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("device_transform: a uint8 (height, width, 3) image expected")
    h, w = int(t.shape[0]), int(t.shape[1])
    s = image_size
    if w <= h:
        nw, nh = s, int(s * h / w)
    else:
        nw, nh = int(s * w / h), s
    if nw < crop_size or nh < crop_size:
        raise ValueError("device_transform: the resized image is smaller than the crop")
    left, top = int(round((nw - crop_size) / 2.0)), int(round((nh - crop_size) / 2.0))
    key = (h, w, nw, nh, left, top, crop_size, str(device))
    tabs = _TABLES.get(key)
    if tabs is None:
        ht, hk = resample_table(w, nw, left, crop_size)
        vt, vk = resample_table(h, nh, top, crop_size)
        if len(_TABLES) > 64:
            _TABLES.clear()
        tabs = (torch.from_numpy(ht).to(device), hk, torch.from_numpy(vt).to(device), vk)
        _TABLES[key] = tabs
    src = t.contiguous().to(device, non_blocking=True)
    outs = list(out) if out is not None else [None, None, None]
    want = (want01, mean_a is not None, mean_b is not None)
    for i in range(3):
        if want[i] and outs[i] is None:
            outs[i] = torch.empty(3, crop_size, crop_size, dtype=torch.float32, device=device)
        if not want[i]:
            outs[i] = None
    f3 = ctypes.c_float * 3
    arg = lambda v: f3(*[float(x) for x in v]) if v is not None else None
    ptr = lambda x: 0 if x is None else x.data_ptr()
    with torch.cuda.device(device):
        _lib.check(_lib.lib.sempyr_resize_crop_normalize(src.data_ptr(), h, w, crop_size, crop_size, tabs[0].data_ptr(), tabs[1],
                                                         tabs[2].data_ptr(), tabs[3], arg(mean_a), arg(std_a), ptr(outs[1]),
                                                         arg(mean_b), arg(std_b), ptr(outs[2]), ptr(outs[0]),
                                                         torch._C._cuda_getCurrentRawStream(torch.device(device).index or 0)),
                   "sempyr_resize_crop_normalize")
    return tuple(outs)


def raw_collate(batch):
    """collate_fn for ImageFolder(raw=True): images of different sizes stay a list"""
    return [b[0] for b in batch], torch.tensor([b[1] for b in batch])


def device_batch(raw_images, device, image_size=256, crop_size=256, mean_a=None, std_a=None, mean_b=None, std_b=None):
    """device_transform of a list of decoded images into batch tensors: ((B,3,crop,crop) in [0,1], copy a, copy b)."""
    B = len(raw_images)
    mk = lambda on: torch.empty(B, 3, crop_size, crop_size, dtype=torch.float32, device=device) if on else None
    x01, xa, xb = mk(True), mk(mean_a is not None), mk(mean_b is not None)
    for i, img in enumerate(raw_images):
        device_transform(img, device, image_size, crop_size, mean_a, std_a, mean_b, std_b,
                         out=(x01[i], None if xa is None else xa[i], None if xb is None else xb[i]))
    return x01, xa, xb


def make_loader(dataset, batch_size, shuffle=True, num_workers=4, drop_last=True, rank=0, world_size=1):
    sampler = None
    if world_size > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, world_size, rank, shuffle=shuffle, drop_last=drop_last)
        shuffle = False
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, num_workers=num_workers,
                                       pin_memory=True, drop_last=drop_last,
                                       collate_fn=raw_collate if getattr(dataset, 'raw', False) else None)


def make_grid(tensor, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """torchvision.utils.make_grid semantics for a (B,C,H,W) batch: optional global min-max normalisation,
    `nrow` images per row, `padding` pixels between/around."""
    t = tensor.detach().float().cpu()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.size(1) == 1:
        t = t.repeat(1, 3, 1, 1)
    if normalize:
        lo, hi = float(t.min()), float(t.max())
        t = (t.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
    B, C, H, W = t.shape
    xmaps = min(nrow, B)
    ymaps = int(np.ceil(B / xmaps))
    hh, ww = H + padding, W + padding
    grid = t.new_full((C, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= B:
                break
            grid[:, y * hh + padding:y * hh + padding + H, x * ww + padding:x * ww + padding + W] = t[k]
            k += 1
    return grid


def save_image(grid, path):
    """torchvision.utils.save_image for an already-assembled (C,H,W) grid in [0,1]."""
    from PIL import Image
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)


# ---- TensorBoard event files, written directly (tensorboardX, which train_model.py:14,290 uses, is not installed) ----------
def _crc32c_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def crc32c(data):
    """CRC-32C (Castagnoli), the checksum of the TFRecord framing."""
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_bytes(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time, step=None, file_version=None, tag=None, value=None):
    """a serialized tensorflow.Event: wall_time (1, double), step (2, int64), file_version (3, string) or
    summary (5) { value (1) { tag (1, string), simple_value (2, float) } }"""
    import struct
    msg = b'\x09' + struct.pack('<d', wall_time)
    if step is not None:
        msg += _varint((2 << 3) | 0) + _varint(step & 0xFFFFFFFFFFFFFFFF)
    if file_version is not None:
        msg += _pb_bytes(3, file_version.encode())
    if tag is not None:
        val = _pb_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack('<f', value)
        msg += _pb_bytes(5, _pb_bytes(1, val))
    return msg


def _record(data):
    import struct
    head = struct.pack('<Q', len(data))
    return head + struct.pack('<I', _masked_crc(head)) + data + struct.pack('<I', _masked_crc(data))


def read_event_scalars(path):
    """[(step, tag, value)] of an event file written by SummaryWriter (checks both checksums of every record)."""
    import struct
    out = []
    with open(path, 'rb') as f:
        raw = f.read()
    pos = 0

    def varint(buf, i):
        n = shift = 0
        while True:
            b = buf[i]; i += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, i
    while pos < len(raw):
        head = raw[pos:pos + 8]
        (n,) = struct.unpack('<Q', head)
        assert struct.unpack('<I', raw[pos + 8:pos + 12])[0] == _masked_crc(head), 'length checksum'
        data = raw[pos + 12:pos + 12 + n]
        assert struct.unpack('<I', raw[pos + 12 + n:pos + 16 + n])[0] == _masked_crc(data), 'data checksum'
        pos += 16 + n
        i, step, summary = 0, 0, None
        while i < len(data):
            key, i = varint(data, i)
            field, wt = key >> 3, key & 7
            if wt == 1:
                i += 8
            elif wt == 5:
                i += 4
            elif wt == 0:
                v, i = varint(data, i)
                if field == 2:
                    step = v
            else:
                ln, i = varint(data, i)
                if field == 5:
                    summary = data[i:i + ln]
                i += ln
        if summary is not None:
            _, j = varint(summary, 0)
            ln, j = varint(summary, j)
            val = summary[j:j + ln]
            k, tag, x = 0, None, None
            while k < len(val):
                key, k = varint(val, k)
                if key == ((1 << 3) | 2):
                    ln2, k = varint(val, k)
                    tag = val[k:k + ln2].decode(); k += ln2
                elif key == ((2 << 3) | 5):
                    (x,) = struct.unpack('<f', val[k:k + 4]); k += 4
            out.append((step, tag, x))
    return out


class SummaryWriter:
    """tensorboardX.SummaryWriter's add_scalar / close (train_model.py:290,333,342), writing a TensorBoard event file
    (`events.out.tfevents.<time>.<host>`: TFRecord framing with masked CRC-32C, hand-encoded Event / Summary protos) that
    TensorBoard reads like tensorboardX's."""

    def __init__(self, logdir):
        import socket
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._f.write(_record(_event(time.time(), file_version='brain.Event:2')))
        self._f.flush()

    def add_scalar(self, tag, value, global_step=None):
        self._f.write(_record(_event(time.time(), step=int(global_step or 0), tag=tag, value=float(value))))
        self._f.flush()

    def close(self):
        if self._f:
            self._f.close()
            self._f = None
