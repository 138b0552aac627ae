"""Drop-in for the part of the reference's utility_functions.py that is on the hot path: `spectrum_fast`
(utility_functions.py:129-155, called at model.py:562), the STFT magnitude / phase feature extractor.

Same name, same arguments, same result layout.  The transform runs on the GPU (csrc/stft.hip through
seld_stft_magphase_ex); there is no CPU path -- a missing device or library raises.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


def _window_values(window, nperseg, device):
    """Window divided by its sum (scipy.signal.stft's scaling='spectrum'), or None for the kernel's built-in periodic
    Hamming.  Other windows need scipy.signal.get_window (host side, a constant table of nperseg values)."""
    if isinstance(window, str) and window == 'hamming':
        return None
    if isinstance(window, (str, tuple)):
        from scipy.signal import get_window
        w = get_window(window, nperseg)
    else:
        w = np.asarray(window, dtype=np.float64)
        if w.shape != (nperseg,):
            raise ValueError('window must have length of nperseg')
    return torch.from_numpy((w / w.sum()).astype(np.float32)).to(device)


def spectrum_fast(x, nperseg=512, noverlap=128, window='hamming', cut_dc=True, output_phase=True,
                  cut_last_timeframe=True):
    '''
    Compute magnitude (and phase) spectra of a multichannel signal -- utility_functions.py:129-155:
    scipy.signal.stft(x, window, nperseg, noverlap) -> |Z| [-> concatenated with angle(Z) on the channel axis]
    [-> DC bin dropped] [-> last frame dropped].

    x: (channels, samples).  A numpy array gives a numpy array of the reference's dtype -- float32 for float32 input
    (scipy's stft returns complex64 then), float64 for anything else; the transform itself is computed in float32 on
    the device -- and a torch tensor gives a float32 tensor on the GPU.
    Returns (channels or 2*channels, nperseg/2 + 1 - cut_dc, frames - cut_last_timeframe).

    Batched input (..., channels, samples) follows the reference literally: the transform runs over the last axis, phase
    is concatenated on axis -3, and the two cuts are the reference's `output[:, 1:, :]` / `output[:, :, :-1]` -- which on
    a batched array act on axes 1 and 2 (channels and frequency), not on the DC bin and the last frame.
    '''
    is_numpy = not torch.is_tensor(x)
    if is_numpy:
        x = np.asarray(x)
        out_dtype = np.float32 if x.dtype == np.float32 else np.float64
    if (x.ndim if is_numpy else x.dim()) > 2:
        lead, C = tuple(x.shape[:-2]), x.shape[-2]
        flat = x.reshape((-1, x.shape[-1]))
        if is_numpy:
            flat = torch.as_tensor(np.ascontiguousarray(flat, dtype=np.float32))
        rows = flat.shape[0]
        both = spectrum_fast(flat, nperseg, noverlap, window, False, output_phase, False)   # [|Z| rows ; angle rows]
        output = both[:rows].reshape(lead + (C,) + tuple(both.shape[1:]))
        if output_phase:
            output = torch.cat((output, both[rows:].reshape(output.shape)), dim=-3)
        if cut_dc:
            output = output[:, 1:, :]
        if cut_last_timeframe:
            output = output[:, :, :-1]
        return output.cpu().numpy().astype(out_dtype) if is_numpy else output
    t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)) if is_numpy else x
    if t.dim() != 2:
        raise ValueError(f"spectrum_fast expects (channels, samples), got shape {tuple(t.shape)}")
    if not torch.cuda.is_available():
        raise L.SeldHipError("spectrum_fast: no HIP device (this package has no CPU path)")
    dev = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    t = t.to(device=dev, dtype=torch.float32).contiguous()
    C, n = t.shape
    nperseg, noverlap = int(nperseg), int(noverlap)
    if noverlap >= nperseg:
        raise ValueError('noverlap must be less than nperseg.')          # scipy's message
    lib = L.lib()
    frames = lib.seld_stft_frames_ex(n, nperseg, noverlap, int(bool(cut_last_timeframe)))
    if frames <= 0:
        raise L.SeldHipError("spectrum_fast: invalid segment parameters")
    bins = nperseg // 2 + 1 - int(bool(cut_dc))
    win = _window_values(window, nperseg, dev)
    out = torch.empty(((2 if output_phase else 1) * C, bins, frames), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        L.check(lib.seld_stft_magphase_ex(L.ptr(t), C, n, nperseg, noverlap, int(bool(output_phase)), int(bool(cut_dc)),
                                          int(bool(cut_last_timeframe)), L.ptr(win), L.ptr(out), L.current_stream()),
                "seld_stft_magphase_ex")
    if is_numpy:
        return out.cpu().numpy().astype(out_dtype)
    return out
