"""Evaluation harness around the hot path ("next" row f1 of SURVEY.md §8f): what ``run_test.py`` needs to turn a folder
of clips + a checkpoint into the PSNR/SSIM table of the paper.

* ``ValFolderDataset``      <-> /root/reference/Experimental_root/data/video_dali_dataset.py:199-249 and
                                 data/utils_common.py:79-192 (numeric filename sort, <= N frames, RGB in [0,1],
                                 CPU AWGN drawn with ``torch.FloatTensor(size).normal_(0, sigma/255)`` from the GLOBAL
                                 torch RNG -- so the realisation depends on the seed and the iteration order exactly as in
                                 the reference --, constant noise map, ``blind`` drops it)
* ``tensor2img``            <-> /root/reference/BasicSR/basicsr/utils/img_util.py:38-94 (clamp, x255, ROUND, RGB->BGR)
* ``calculate_psnr`` / ``calculate_psnr_float`` / ``calculate_ssim``
                            <-> /root/reference/BasicSR/basicsr/metrics/psnr_ssim.py:9-168
* ``evaluate``              <-> the metric part of DenoisingModel.nondist_validation (denoising_model.py:215-367):
                                 per-frame metrics, per-folder mean, mean over folders.

Image decoding uses PIL (cv2 is not available here); cv2.imread's BGR + cvtColor(BGR2RGB) equals PIL's RGB.
PSNR is pinned to the reference by tests/golden/g9_psnr.npz (the reference's own metric functions).  SSIM is NOT pinned
to cv2: the reference's ``_ssim`` needs ``cv2.getGaussianKernel`` / ``cv2.filter2D``, cv2 is absent from the build
container, and golden g12 runs the reference's formula over numpy STAND-INS for those two calls -- it pins the formula
and this restatement (11x11 gaussian, sigma 1.5, 'valid' window; also tested against an independent direct window sum),
not OpenCV's arithmetic.
"""
import glob
import os

import numpy as np
import torch

from .registry import register_dataset

IMAGETYPES = ('*.bmp', '*.png', '*.jpg', '*.jpeg', '*.tif')


def image_names(seq_dir):
    files = []
    for typ in IMAGETYPES:
        files.extend(glob.glob(os.path.join(seq_dir, typ)))
    # the reference sorts by the integer formed by ALL digits of the path (utils_common.py:95)
    files.sort(key=lambda f: int(''.join(filter(str.isdigit, f)) or 0))
    return files


def open_sequence(seq_dir, max_num_fr=100):
    """[F,3,H,W] float32 RGB in [0,1]."""
    from PIL import Image
    frames = []
    for path in image_names(seq_dir)[:max_num_fr]:
        with Image.open(path) as im:
            frames.append(np.asarray(im.convert("RGB"), dtype=np.uint8).transpose(2, 0, 1))
    if not frames:
        raise FileNotFoundError("no images in %s" % seq_dir)
    return np.float32(np.stack(frames, 0) / 255.)


@register_dataset
class ValFolderDataset:
    """opt keys like the reference: valsetdir, num_validation_frames, valnoisestd, [scene_name], [blind], [name].
    In DATASET_REGISTRY (video_dali_dataset.py:199-200), so ``basicsr.data.build_dataset`` finds it; an instance is what
    ``DenoisingModel.nondist_validation`` reads through ``dataloader.dataset`` (opt['name'], base_folder, num_frames)."""

    def __init__(self, opt, device=None):
        self.opt = opt
        self.device = device if device is not None else (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        root = opt['valsetdir']
        dirs = sorted(p for p in glob.glob(os.path.join(root, '*')) if os.path.isdir(p))
        if opt.get('scene_name') is not None:
            dirs = [d for d in dirs if opt['scene_name'] in d]
        self.seqs_dirs = dirs
        self.base_folder = [os.path.basename(d) for d in dirs]
        self.num_input_frames = opt['num_validation_frames']
        self.num_frames = [min(len(image_names(d)), self.num_input_frames) for d in dirs]

    def __len__(self):
        return len(self.seqs_dirs)

    def __getitem__(self, index):
        gt = torch.from_numpy(open_sequence(self.seqs_dirs[index], self.num_input_frames))[None]
        n, f, _, h, w = gt.shape
        sigma = self.opt['valnoisestd'] / 255.0
        noise = torch.FloatTensor(gt.size()).normal_(mean=0, std=sigma)          # global CPU RNG, like the reference
        item = {'gt': gt, 'lq': (gt + noise).to(self.device),
                'noise_map': torch.full((n, f, 1, h, w), sigma, dtype=torch.float32, device=self.device),
                'folder': self.base_folder[index], 'index': index}
        if self.opt.get('blind', False):
            item.pop('noise_map')
        return item


# ------------------------------------------------------------------------------------------------- metrics
def tensor2img(t, rgb2bgr=True, min_max=(0, 1)):
    """[3,H,W] (or [1,3,H,W]) float tensor in RGB -> uint8 HWC (BGR by default), rounded like the reference."""
    a = t.squeeze(0).float().detach().cpu().clamp(*min_max)
    a = ((a - min_max[0]) / (min_max[1] - min_max[0])).numpy()
    if a.ndim == 3:
        a = a.transpose(1, 2, 0)
        if a.shape[2] == 1:
            a = a[..., 0]
        elif rgb2bgr:
            a = a[..., ::-1]
    return (a * 255.0).round().astype(np.uint8)


def _hwc(img, input_order):
    if img.ndim == 2:
        return img[..., None]
    return img.transpose(1, 2, 0) if input_order == 'CHW' else img


def _crop(img, b):
    return img if b == 0 else img[b:-b, b:-b, ...]


def _no_y(test_y_channel):
    if test_y_channel:
        raise NotImplementedError("test_y_channel=True is not used by the BSVD configs")


def calculate_psnr(img, img2, crop_border, input_order='HWC', test_y_channel=False):
    """uint8-domain PSNR, range [0,255]."""
    _no_y(test_y_channel)
    a = _crop(_hwc(np.asarray(img), input_order).astype(np.float64), crop_border)
    b = _crop(_hwc(np.asarray(img2), input_order).astype(np.float64), crop_border)
    mse = np.mean((a - b) ** 2)
    return float('inf') if mse == 0 else float(20. * np.log10(255. / np.sqrt(mse)))


def calculate_psnr_float(img_float, img2_float, crop_border, input_order='CHW', test_y_channel=False):
    """float-domain PSNR, range [0,1] (the BSVD authors' addition)."""
    _no_y(test_y_channel)
    a = _crop(_hwc(img_float.detach().cpu().numpy(), input_order), crop_border)
    b = _crop(_hwc(img2_float.detach().cpu().numpy(), input_order), crop_border)
    mse = np.mean((a - b) ** 2)
    return float('inf') if mse == 0 else float(-10 * np.log10(mse))


def _gauss_kernel(n=11, sigma=1.5):
    x = np.arange(n, dtype=np.float64) - (n - 1) / 2
    k = np.exp(-(x ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def _filt_valid(a, k):
    """separable 'valid' correlation with the symmetric kernel k along both axes"""
    n = len(k)
    h, w = a.shape
    tmp = np.zeros((h - n + 1, w), dtype=np.float64)
    for i in range(n):
        tmp += k[i] * a[i:i + h - n + 1, :]
    out = np.zeros((h - n + 1, w - n + 1), dtype=np.float64)
    for i in range(n):
        out += k[i] * tmp[:, i:i + w - n + 1]
    return out


def _ssim(a, b):
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = a.astype(np.float64), b.astype(np.float64)
    k = _gauss_kernel()
    mu1, mu2 = _filt_valid(a, k), _filt_valid(b, k)
    s1 = _filt_valid(a * a, k) - mu1 ** 2
    s2 = _filt_valid(b * b, k) - mu2 ** 2
    s12 = _filt_valid(a * b, k) - mu1 * mu2
    return float((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean())


def calculate_ssim(img, img2, crop_border, input_order='HWC', test_y_channel=False):
    _no_y(test_y_channel)
    a = _crop(_hwc(np.asarray(img), input_order).astype(np.float64), crop_border)
    b = _crop(_hwc(np.asarray(img2), input_order).astype(np.float64), crop_border)
    return float(np.mean([_ssim(a[..., i], b[..., i]) for i in range(a.shape[2])]))


METRICS = {"calculate_psnr": calculate_psnr, "calculate_psnr_float": calculate_psnr_float, "calculate_ssim": calculate_ssim}


def frame_metrics(result, gt, metrics_opt):
    """{name: value} of one frame pair ([3,H,W] float tensors): 'float' metrics on the tensors, the others on the uint8
    BGR images ``tensor2img`` makes of them (denoising_model.py:275-310, basicsr/metrics/__init__.py calculate_metric)."""
    out = {}
    imgs = None
    for name, mo in metrics_opt.items():
        mo = dict(mo)
        typ = mo.pop('type')
        fn = METRICS[typ]
        if 'float' in typ:
            out[name] = fn(result, gt, **mo)
        else:
            if imgs is None:
                imgs = (tensor2img(result), tensor2img(gt))
            out[name] = fn(imgs[0], imgs[1], **mo)
    return out


def imwrite(img_bgr, path):
    """basicsr.utils.imwrite (cv2.imwrite of a BGR uint8 image, parent directory created) on PIL."""
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(img_bgr[..., ::-1] if img_bgr.ndim == 3 else img_bgr).save(path)


def evaluate(model, dataset, metrics_opt, per_frame=None, save_img_dir=None, run_name="bsvd"):
    """model: bsvd_amd.DenoisingModel; metrics_opt: {name: {type: calculate_psnr, crop_border: 2}, ...} as in
    options/test/bsvd_c64.yml:116-123.  Returns ({folder: {metric: mean}}, {metric: mean over folders}).
    per_frame (a dict) receives {folder: {metric: [value per frame]}} -- what the reference writes to one CSV per folder
    (denoising_model.py:335-345); save_img_dir writes the denoised frames as <dir>/<folder>/<idx:08d>_<run_name>.png
    (uint8 BGR->RGB, the reference's save_img naming, :295-299)."""
    per_folder = {}
    for i in range(len(dataset)):
        item = dataset[i]
        model.feed_data(item)
        model.test()
        vis = model.get_current_visuals()
        res, gt = vis['result'][0], vis['gt'][0] if vis['gt'].dim() == 5 else vis['gt']
        acc = {k: [] for k in metrics_opt}
        for f in range(res.shape[0]):
            for name, v in frame_metrics(res[f], gt[f], metrics_opt).items():
                acc[name].append(v)
        per_folder[item['folder']] = {k: float(np.mean(v)) for k, v in acc.items()}
        if per_frame is not None:
            per_frame[item['folder']] = acc
        if save_img_dir is not None:
            for f in range(res.shape[0]):
                imwrite(tensor2img(res[f]), os.path.join(save_img_dir, item['folder'], "%08d_%s.png" % (f, run_name)))
    total = {k: float(np.mean([v[k] for v in per_folder.values()])) for k in metrics_opt} if per_folder else {}
    return per_folder, total
