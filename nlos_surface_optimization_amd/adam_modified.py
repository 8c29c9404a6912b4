"""Drop-in for the reference's `adam_modified.Adam_Modified` (exp_bunny/adam_modified.py).

Same constructor, parameter groups and state keys (`step`, `exp_avg`, `exp_avg_sq`,
`max_exp_avg_sq`); the update itself -- Adam with ONE denominator per row, the mean of
sqrt(exp_avg_sq) + eps over the row (adam_modified.py:96-107) -- runs as one HIP kernel per
parameter on the tensor's device, so vertices, gradient and optimiser state stay in HBM.
`p.grad` may be float32 or the renderer's float64 output (narrowed in the kernel exactly like the
reference's `torch.from_numpy(grad).float()`, exp_bunny/test.py:212-213).  Parameters must be
2-D float32 HIP tensors with at most 8 columns; there is no CPU path.
"""
import ctypes

import torch
from torch.optim.optimizer import Optimizer

from . import _lib


class Adam_Modified(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        # same argument checks and messages as the reference class (exp_bunny/adam_modified.py:33-40)
        b1, b2 = betas
        for bad, what, val in ((lr < 0.0, "learning rate", lr), (eps < 0.0, "epsilon value", eps),
                               (not 0.0 <= b1 < 1.0, "beta parameter at index 0", b1),
                               (not 0.0 <= b2 < 1.0, "beta parameter at index 1", b2)):
            if bad:
                raise ValueError("Invalid %s: %s" % (what, val))
        super().__init__(params, {"lr": lr, "betas": (b1, b2), "eps": eps, "weight_decay": weight_decay,
                                  "amsgrad": amsgrad})
        self._ctx = {}          # one render context (scratch owner) per device

    def __setstate__(self, state):
        super().__setstate__(state)
        for g in self.param_groups:
            g.setdefault("amsgrad", False)

    def _handle(self, device):
        h = self._ctx.get(device.index)
        if h is None:
            h = ctypes.c_void_p()
            _lib.check(_lib.lib().nlos_ctx_create(device.index or 0, ctypes.byref(h)), "nlos_ctx_create")
            self._ctx[device.index] = h
        return h

    def __del__(self):
        try:
            for h in self._ctx.values():
                _lib.lib().nlos_ctx_destroy(h)
        except Exception:
            pass

    @staticmethod
    def assign_grad(p, grad):
        """p.grad = grad, also for the renderer's float64 gradient on a float32 parameter (recent
        torch versions refuse a gradient whose dtype differs unless `grad_dtype` is relaxed)."""
        if grad.dtype != p.dtype and hasattr(p, "grad_dtype"):
            p.grad_dtype = None
        p.grad = grad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad.data
                if grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                if not p.is_cuda:
                    raise _lib.NlosError("Adam_Modified: parameters must live on an AMD GPU (no CPU fallback)")
                if p.dtype != torch.float32 or p.dim() != 2 or p.shape[1] > 8 or not p.is_contiguous():
                    raise ValueError("Adam_Modified: parameters must be contiguous float32 [rows, cols <= 8]")
                if grad.shape != p.shape or grad.device != p.device or not grad.is_contiguous():
                    raise ValueError("Adam_Modified: grad must be a contiguous tensor of the parameter's shape and device")
                if grad.dtype not in (torch.float32, torch.float64):
                    raise ValueError("Adam_Modified: grad must be float32 or float64")
                state = self.state[p]
                if not state:
                    state.update(step=0, exp_avg=torch.zeros_like(p.data), exp_avg_sq=torch.zeros_like(p.data))
                    if group['amsgrad']:
                        state['max_exp_avg_sq'] = torch.zeros_like(p.data)
                state['step'] += 1
                beta1, beta2 = group['betas']
                g64 = ctypes.c_void_p(grad.data_ptr()) if grad.dtype == torch.float64 else None
                g32 = ctypes.c_void_p(grad.data_ptr()) if grad.dtype == torch.float32 else None
                mx = ctypes.c_void_p(state['max_exp_avg_sq'].data_ptr()) if group['amsgrad'] else None
                stream = torch.cuda.current_stream(p.device).cuda_stream
                with torch.cuda.device(p.device):
                    rc = _lib.lib().nlos_adam_modified_step(
                        self._handle(p.device), ctypes.c_void_p(p.data.data_ptr()), g64, g32,
                        ctypes.c_void_p(state['exp_avg'].data_ptr()), ctypes.c_void_p(state['exp_avg_sq'].data_ptr()),
                        mx, None, p.shape[0], p.shape[1], int(state['step']), float(group['lr']), float(beta1),
                        float(beta2), float(group['eps']), float(group['weight_decay']), ctypes.c_void_p(stream))
                _lib.check(rc, "nlos_adam_modified_step")
        return loss
