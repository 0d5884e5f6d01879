"""The on-disk formats either side of the hot path (row f4; modules/dataset/dataset.py of the reference): which
models belong to which split, the camera of every rendering, and how an RGBA rendering becomes the network input
and the GT silhouette.  Host-side text / tensor handling only — the dataset class itself (file discovery,
augmentation, kaolin mesh sampling) is out of scope."""
import torch

DIST_SCALE = 1.754                       # dataset.py:147: the stored camera distance is scaled by this factor
IMAGENET_MEAN = (0.485, 0.456, 0.406)    # dataset.py:126
IMAGENET_STD = (0.229, 0.224, 0.225)


def parse_split_csv(text: str) -> dict:
    """ShapeNet's split.csv (`id,synsetId,subSynsetId,modelId,split`, dataset.py:98-106) ->
    {'train': [(synsetId, modelId), ...], 'test': [...]}; `val` rows are added to `train` after the train rows,
    the header and malformed rows are skipped."""
    rows = {'train': [], 'val': [], 'test': []}
    for line in text.splitlines():
        f = line.strip().split(',')
        if len(f) >= 5 and f[-1] in rows:
            rows[f[-1]].append((f[1], f[3]))
    return {'train': rows['train'] + rows['val'], 'test': rows['test']}


def parse_rendering_metadata(text: str):
    """rendering_metadata.txt of the ShapeNet renderings (one `azim elev 0 dist 25` line per view,
    dataset.py:141-151) -> (azims, elevs, dists) as lists of floats, dists already multiplied by 1.754.
    (The reference's pattern has no sign: it would read `-10` as `10`; signs are kept here.)"""
    azims, elevs, dists = [], [], []
    for line in text.splitlines():
        t = line.split()
        if len(t) >= 5 and t[2] == '0' and t[4] == '25':
            azims.append(float(t[0]))
            elevs.append(float(t[1]))
            dists.append(float(t[3]) * DIST_SCALE)
    return azims, elevs, dists


def split_rgba(img: torch.Tensor, normalize: bool = False):
    """An RGBA rendering as a (4,H,W) tensor in [0,1] (what ToTensor gives, dataset.py:116-117) ->
    rgb (3,H,W) and silhouette (1,H,W) = the alpha channel (dataset.py:123); optional ImageNet normalisation of
    the rgb part (dataset.py:125-126).  Batched (B,4,H,W) input works the same way."""
    assert img.size(-3) == 4
    rgb, sil = img[..., :3, :, :], img[..., 3:4, :, :]
    if normalize:
        mean = torch.tensor(IMAGENET_MEAN, dtype=rgb.dtype, device=rgb.device).view(3, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=rgb.dtype, device=rgb.device).view(3, 1, 1)
        rgb = (rgb - mean) / std
    return rgb, sil
