"""Host-side I/O the CLIs need and torchvision / tensorboardX used to provide (neither is installed
here): an ImageFolder with Resize(256)+CenterCrop(256)+ToTensor, a sharded
loader, make_grid / save_image, and a scalar writer.  None of this is on the timed path."""
import os
import time

import numpy as np
import torch

IMG_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


class ImageFolder(torch.utils.data.Dataset):
    """root/<class>/<image> -> (float tensor (3,crop,crop) in [0,1], class index); same transform chain as
    train_model.py:233-238 (shorter side -> image_size, bilinear with antialias; center crop; ToTensor)."""

    def __init__(self, root, image_size=256, crop_size=256):
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


def make_loader(dataset, batch_size, shuffle=True, num_workers=4, drop_last=True, rank=0, world_size=1):
    sampler = None
    if world_size > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, world_size, rank, shuffle=shuffle, drop_last=drop_last)
        shuffle = False
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, num_workers=num_workers,
                                       pin_memory=True, drop_last=drop_last)


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
