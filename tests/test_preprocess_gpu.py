"""Input pipeline on the device: sempyr_resize_crop_normalize against the host chain of
imageio.ImageFolder (PIL Resize(256) bilinear + antialias, CenterCrop(256), ToTensor) followed by the two Normalize maps
(train_model.py:233-247,308-309) -- bit for bit, for images that are down-scaled, up-scaled, portrait, landscape and already
256 x 256."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_chain(arr, s=256, crop=256):
    from PIL import Image
    h, w = arr.shape[:2]
    nw, nh = (s, int(s * h / w)) if w <= h else (int(s * w / h), s)
    img = Image.fromarray(arr).resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - crop) / 2.0)), int(round((nh - crop) / 2.0))
    img = img.crop((left, top, left + crop, top + crop))
    return torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).float().div(255)


@pytest.mark.parametrize("size", [(300, 400), (512, 512), (256, 256), (700, 333), (200, 260), (1031, 777), (256, 1024)])
def test_device_transform_matches_pil_chain(dev, size):
    from semantic_pyramid_for_image_generation_amd import imageio
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    h, w = size
    arr = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    want01 = _host_chain(arr).to(dev)
    mean_c, std_c = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    x01, xc, xd = imageio.device_transform(arr, dev, mean_a=mean_c, std_a=std_c, mean_b=[0.5] * 3, std_b=[0.5] * 3)
    assert torch.equal(x01, want01)
    assert torch.equal(xc, tm.Normalize(mean_c, std_c)(want01[None])[0])
    assert torch.equal(xd, tm.Normalize([0.5] * 3, [0.5] * 3)(want01[None])[0])
    only_d = imageio.device_transform(arr, dev, mean_b=[0.5] * 3, std_b=[0.5] * 3, want01=False)
    assert only_d[0] is None and only_d[1] is None and torch.equal(only_d[2], xd)


def test_device_batch_and_raw_folder(dev, tmp_path):
    """ImageFolder(raw=True) + raw_collate + device_batch == ImageFolder() + default collate, on a folder of mixed sizes"""
    from PIL import Image
    from semantic_pyramid_for_image_generation_amd import imageio
    rng = np.random.default_rng(3)
    (tmp_path / 'a').mkdir(); (tmp_path / 'b').mkdir()
    for i, (h, w) in enumerate([(260, 300), (400, 256), (256, 256), (513, 777)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / ('a' if i % 2 else 'b') / f'{i}.png')
    host = imageio.ImageFolder(str(tmp_path))
    raw = imageio.ImageFolder(str(tmp_path), raw=True)
    lh = imageio.make_loader(host, 4, shuffle=False, num_workers=0, drop_last=False)
    lr = imageio.make_loader(raw, 4, shuffle=False, num_workers=0, drop_last=False)
    (xh, th), (xr, tr) = next(iter(lh)), next(iter(lr))
    assert isinstance(xr, list) and len(xr) == 4 and torch.equal(th, tr)
    got = imageio.batch_images(xr, dev)
    assert torch.equal(got, xh.to(dev))
