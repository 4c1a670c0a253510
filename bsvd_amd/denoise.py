"""Callers of the hot path, call-compatible with the reference's inference harness:

* ``temp_denoise`` / ``denoise_seq``  <->  /root/reference/Experimental_root/models/validation_seq_infer.py:10-100
* ``DenoisingModel`` (test path only) <->  /root/reference/Experimental_root/models/denoising_model.py:16-190
  (``feed_data`` :91, ``padding_input`` :133, ``crop_output`` :161, ``test`` :170, ``get_current_visuals`` :369)

Semantics reproduced (pinned by tests/golden/g8_pad_crop_clamp.npz): H and W are padded on the right/bottom to
multiples of 4 with reflection; the noise map is rebuilt as a constant from its first element; with
``temp_psz == -1`` the whole clip goes through the network in ONE call; outputs are clamped to [0, 1]; the pad is
cropped again.  For ``temp_psz > 0`` the clip is cut into independent segments with a mirrored tail, as the
reference does for a ``BSVD`` network (whose buffers are reset per call).  Training, losses, EMA, logging and image
dumping of the reference class are out of scope.
"""
import torch
import torch.nn.functional as F

from . import global_queue_buffer
from .registry import _ENGINE, build_network, register_model


def temp_denoise(model, noisyframe, sigma_noise, device=None):
    """noisyframe [F,C,H,W] in [0,1]; sigma_noise [F,1,H,W] constant map or None -> clamped [F,C_out,H,W]."""
    n, _, h, w = noisyframe.shape
    clip = noisyframe[None]
    if sigma_noise is not None:
        first = sigma_noise[0, 0, 0, 0]
        if abs(float(sigma_noise.float().mean()) - float(first)) >= 1e-5:
            raise AssertionError("the noise map must be constant (validation_seq_infer.py:19)")
        sigma = torch.full((1, n, 1, h, w), float(first), dtype=noisyframe.dtype, device=noisyframe.device)
        out = model(clip, noise_map=sigma)[0]
    else:
        out = model(clip)[0]
    out = torch.clamp(out, 0.0, 1.0)
    if out.is_cuda:
        torch.cuda.synchronize(out.device)
    return out if device is None else out.to(device)


def denoise_seq(seq, noise_map, temp_psz, model_temporal, future_buffer_len=0):
    """seq [T,C,H,W] -> denoised [T,C,H,W] on seq's device (float32)."""
    dev = next(model_temporal.parameters()).device
    T, C, H, W = seq.shape
    if temp_psz == -1:
        temp_psz = T                                   # BSVD: the whole video in a single forward
    out = torch.empty((T, C, H, W), dtype=torch.float32, device=seq.device)
    nseg = T // temp_psz
    # MIMO state (only TSN reads it; BSVD ignores it): per-layer past slices travel from segment to segment
    global_queue_buffer._init(future_buffer_len)
    try:
        for i in range(nseg):
            global_queue_buffer.set_batch_index(i)
            a, b = i * temp_psz, (i + 1) * temp_psz
            b_in = b + future_buffer_len
            if b_in > T:                        # no look-ahead available for the last full segment
                b_in = b
                global_queue_buffer.set_future_buffer_length(0)
            res = temp_denoise(model_temporal, seq[a:b_in].to(dev), noise_map, seq.device)
            out[a:b] = res[:temp_psz]
        global_queue_buffer.set_future_buffer_length(0)
        rest = T - nseg * temp_psz
        if rest > 0:
            # mirror-extend the tail to a full segment, keep the first `rest` outputs (validation_seq_infer.py:75-90);
            # like the reference the segment index is NOT advanced for the tail
            tail = torch.cat((seq[nseg * temp_psz:], torch.flip(seq[-(temp_psz - rest) - 1:-1], dims=[0])))
            res = temp_denoise(model_temporal, tail.to(dev), noise_map, seq.device)
            out[nseg * temp_psz:] = res[:rest]
    finally:
        global_queue_buffer._clean()
    return out


def pad_to_multiple_of_4(frames):
    """[F,C,H,W] -> (padded, padding_list) with padding_list = [0, pad_w, 0, pad_h, 0, 0] (right/bottom reflect)."""
    h, w = frames.shape[-2:]
    pad_h = (4 - h % 4) % 4
    pad_w = (4 - w % 4) % 4
    padded = F.pad(frames, (0, pad_w, 0, pad_h), mode="reflect") if (pad_h or pad_w) else frames
    return padded, [0, pad_w, 0, pad_h, 0, 0]


def crop_padding(output, padding_list):
    """output [1,F,C,H,W]; inverse of pad_to_multiple_of_4."""
    pw1, pw2, ph1, ph2, t1, t2 = padding_list
    _, f, _, h, w = output.shape
    return output[:, t1:f - t2, :, ph1:h - ph2, pw1:w - pw2]


def build_engine_network(net_opt):
    """``build_network`` for the engine's own model class: a stock ``type`` (``BSVD`` / ``TSN``) always means the ENGINE's
    class of that name, also while the reference plug-in holds the name in a shared BasicSR registry (then the engine's
    class is registered as ``<type>_MI355X``, see registry.py); any other type goes through the registry."""
    net_opt = dict(net_opt)
    cls = _ENGINE["arch"].get(net_opt["type"])
    if cls is None:
        return build_network(net_opt)
    net_opt.pop("type")
    return cls(**net_opt)


@register_model
class DenoisingModel:
    """Inference half of the reference's DenoisingModel (denoising_model.py:16-190, 192-378): opt dict in, ``feed_data`` /
    ``test`` / ``get_current_visuals`` and ``validation`` / ``dist_validation`` / ``nondist_validation`` (what
    ``basicsr.test_pipeline`` calls, BasicSR/basicsr/test.py:37-41) out.  ``opt['network_g']`` is splatted into the arch
    constructor exactly like basicsr.archs.build_network does; ``opt['val']['temp_psz']`` (-1 for BSVD) and
    ``future_buffer_len`` are honoured.  The engine is GPU-only: without a HIP device construction raises."""

    @staticmethod
    def _pick_device():
        if not torch.cuda.is_available():
            raise RuntimeError("DenoisingModel needs a HIP device")
        return torch.device("cuda")

    def __init__(self, opt):
        self.opt = opt
        self.is_train = bool(opt.get("is_train", False))
        if self.is_train:
            raise NotImplementedError("bsvd_amd implements the inference path only (training is out of scope)")
        self.device = self._pick_device()
        self.center_frame_only = bool(opt.get("center_frame_only", False))
        self.net_g = build_engine_network(opt["network_g"]).to(self.device)
        popt = opt.get("path") or {}
        if popt.get("pretrain_network_g"):
            self.load_network(self.net_g, popt["pretrain_network_g"], popt.get("strict_load_g", True),
                              popt.get("param_key_g", "params"))
        self.lq = self.gt = self.noise_map = self.output = None

    @staticmethod
    def load_network(net, load_path, strict=True, param_key="params"):
        """BaseModel.load_network (BasicSR/basicsr/models/base_model.py:252-278): picks ``param_key``, strips the
        DataParallel ``module.`` prefix; additionally a TSN-schema file is re-keyed when the target is ``BSVD``."""
        from . import checkpoint
        from .arch import BSVD
        state = torch.load(load_path, map_location="cpu")
        if param_key not in (None, "None") and isinstance(state, dict) and param_key in state:
            state = state[param_key]
        state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
        if isinstance(net, BSVD) and checkpoint.is_tsn_schema(state):
            state = checkpoint.to_bsvd_state(state)
        net.load_state_dict(state, strict=strict)

    def feed_data(self, data):
        self.lq = data["lq"].to(self.device)
        self.noise_map = data["noise_map"].to(self.device) if "noise_map" in data else None
        if "gt" in data:
            self.gt = data["gt"].to(self.device)

    def padding_input(self, frames):
        return pad_to_multiple_of_4(frames)

    def crop_output(self, padding_list):
        self.output = crop_padding(self.output, padding_list)

    def test(self):
        """lq: [F,C,H,W] (or [1,F,C,H,W]) in [0,1]; result in self.output as [1,F,C,H,W]."""
        lq = self.lq[0] if self.lq.dim() == 5 else self.lq
        nm = None
        if self.noise_map is not None:
            nm = self.noise_map[0] if self.noise_map.dim() == 5 else self.noise_map
        self.net_g.eval()
        val = self.opt.get("val") or {}
        with torch.no_grad():
            padded, plist = self.padding_input(lq)
            pnm = self.padding_input(nm)[0] if nm is not None else None
            self.output = denoise_seq(padded, pnm, val.get("temp_psz", -1), self.net_g,
                                      future_buffer_len=val.get("future_buffer_len", 0))[None]
            self.crop_output(plist)

    def get_current_visuals(self):
        out = {"lq": self.lq.detach().cpu(), "result": self.output.detach().cpu()}
        if self.gt is not None:
            out["gt"] = self.gt.detach().cpu()
        return out

    # ---- what basicsr.test_pipeline drives (BasicSR/basicsr/test.py:37-41) ----------------------------------------
    def validation(self, dataloader, current_iter, tb_logger, save_img=False):
        """denoising_model.py:192-209.  ``val.fp16`` put the reference's convs under autocast; the engine's arithmetic is
        fixed by ``network_g.precision`` (split-fp16 3-pass or exact fp32, both fp32-class), so the flag only logs."""
        if self.opt.get("dist", False):
            return self.dist_validation(dataloader, current_iter, tb_logger, save_img)
        if (self.opt.get("val") or {}).get("fp16", False):
            _logger().info("val.fp16 is set: the MI355X engine keeps its own arithmetic (precision=%s)",
                           getattr(self.net_g, "precision", "?"))
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def dist_validation(self, dataloader, current_iter, tb_logger, save_img):
        """denoising_model.py:211-213: rank 0 evaluates, the others return None."""
        if self.opt.get("rank", 0) == 0:
            return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)
        return None

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img):
        """denoising_model.py:215-323: every folder of ``dataloader.dataset`` through feed_data / test, per-frame metrics
        of ``val.metrics`` into ``self.metric_results[folder]`` ([frames, metrics]), optional PNG dump under
        ``path.visualization/<dataset>/<folder>/<idx:08d>_<name>.png``, then the log line / CSVs of
        ``_log_validation_metric_values``.  Returns {metric: mean over folders}."""
        import os
        from . import evaluation
        dataset = dataloader.dataset if hasattr(dataloader, "dataset") else dataloader
        dataset_name = dataset.opt["name"]
        val = self.opt.get("val") or {}
        metrics = val.get("metrics")
        with_metrics = metrics is not None
        if with_metrics:
            self.metric_results = {folder: torch.zeros(dataset.num_frames[i], len(metrics), dtype=torch.float32)
                                   for i, folder in enumerate(dataset.base_folder)}
        total = None
        for i in range(len(dataset)):
            val_data = dataset[i]
            folder = val_data["folder"]
            self.feed_data(val_data)
            with torch.no_grad():
                self.test()
            visuals = self.get_current_visuals()
            self.lq = self.output = self.noise_map = self.gt = None      # "tentative for out of GPU memory" (:259-265)
            res = visuals["result"][0]
            gt = None
            if "gt" in visuals:
                gt = visuals["gt"][0] if visuals["gt"].dim() == 5 else visuals["gt"]
            for idx in range(res.shape[0]):
                if save_img:
                    path = os.path.join(self.opt["path"]["visualization"], dataset_name, folder,
                                        "%08d_%s.png" % (idx, self.opt.get("name", "bsvd")))
                    evaluation.imwrite(evaluation.tensor2img(res[idx]), path)
                if with_metrics and gt is not None:
                    vals = evaluation.frame_metrics(res[idx], gt[idx], metrics)
                    for mi, name in enumerate(metrics):
                        self.metric_results[folder][idx, mi] += vals[name]
            if with_metrics:
                total = self._log_validation_metric_values(current_iter, dataset_name, tb_logger)
        return total

    def _log_validation_metric_values(self, current_iter, dataset_name, tb_logger):
        """denoising_model.py:325-367: per-folder means, mean over folders, the reference's log line, one CSV of per-frame
        values per folder next to the log file (columns ``<folder>_<metric index>``), tensorboard scalars."""
        metrics = list((self.opt.get("val") or {}).get("metrics").keys())
        avg = {folder: t.mean(dim=0) for folder, t in self.metric_results.items()}
        log = _logger()
        base = _log_file(log)
        if base is not None:
            for folder, t in self.metric_results.items():
                with open(base.replace(".log", "%s.csv" % folder), "w") as fh:
                    fh.write("," + ",".join("%s_%d" % (folder, mi) for mi in range(len(metrics))) + "\n")
                    for r in range(t.shape[0]):
                        fh.write("%d," % r + ",".join(str(t[r, mi].numpy()) for mi in range(len(metrics))) + "\n")
        total = {m: sum(float(a[mi]) for a in avg.values()) / max(len(avg), 1) for mi, m in enumerate(metrics)}
        msg = "Validation %s\n" % dataset_name
        for mi, (metric, value) in enumerate(total.items()):
            msg += "\t # %s: %.4f" % (metric, value)
            for folder, a in avg.items():
                msg += "\t # %s: %.4f" % (folder, float(a[mi]))
            msg += "\n"
        log.info(msg)
        if tb_logger:
            for mi, (metric, value) in enumerate(total.items()):
                tb_logger.add_scalar("metrics/%s" % metric, value, current_iter)
                for folder, a in avg.items():
                    tb_logger.add_scalar("metrics/%s/%s" % (metric, folder), float(a[mi]), current_iter)
        return total


def _logger():
    """basicsr.utils.get_root_logger's logger (name 'basicsr'); test_pipeline attaches the stream + file handlers."""
    import logging
    return logging.getLogger("basicsr")


def _log_file(log):
    import logging
    for h in log.handlers:
        if isinstance(h, logging.FileHandler):
            return h.baseFilename
    return None
