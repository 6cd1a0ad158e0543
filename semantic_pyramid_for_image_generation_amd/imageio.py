"""Host-side I/O the CLIs need and torchvision / tensorboardX used to provide (neither is installed
here): an ImageFolder with Resize(256)+CenterCrop(256)+ToTensor, a sharded
loader, make_grid / save_image, and a scalar writer.  None of this is on the timed path."""
import json
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
        arr = np.asarray(img, dtype=np.uint8)
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


class SummaryWriter:
    """Scalar logger with tensorboardX's add_scalar signature.  Uses tensorboardX / torch.utils.tensorboard when
    importable, else appends JSON lines to <logdir>/scalars.jsonl."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self._tb = None
        for mod in ('tensorboardX', 'torch.utils.tensorboard'):
            try:
                self._tb = __import__(mod, fromlist=['SummaryWriter']).SummaryWriter(logdir)
                break
            except Exception:
                continue
        self._f = None if self._tb is not None else open(os.path.join(logdir, 'scalars.jsonl'), 'a')

    def add_scalar(self, tag, value, global_step=None):
        if self._tb is not None:
            self._tb.add_scalar(tag, value, global_step=global_step)
        else:
            self._f.write(json.dumps({'tag': tag, 'value': float(value), 'step': global_step, 'time': time.time()}) + '\n')
            self._f.flush()

    def close(self):
        if self._tb is not None:
            self._tb.close()
        elif self._f:
            self._f.close()
