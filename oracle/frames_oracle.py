"""CPU restatement of the per-frame host work either side of VToonify.forward in the reference's
video loop (style_transfer.py:99-183).  TEST INFRASTRUCTURE ONLY: imported by tests/ (and never by
the product path), like the rest of oracle/.

Parity status: torchvision and cv2 are not installed in this image, so the three library calls
the reference makes here are restated from their documented semantics rather than executed:
  transforms.ToTensor()            uint8 HWC -> float32 CHW, `.div(255)`              (style_transfer.py:58)
  transforms.Normalize(.5, .5)     `tensor.sub_(mean).div_(std)` per channel          (style_transfer.py:59)
  cv2.cvtColor(RGB2BGR / BGR2RGB)  channel reversal of an 8-bit 3-channel image       (style_transfer.py:114, util.py:192)
Everything else (x_p/16, clamp, (x+1)*127.5, astype(uint8)) is the reference's own numpy/torch code
and is restated verbatim.  All arithmetic is float32, one rounding per operation.
"""
import numpy as np


def to_tensor_normalize(frame_rgb: np.ndarray) -> np.ndarray:
    """transform(frame) of style_transfer.py:57-60,160: (H,W,3) uint8 RGB -> (3,H,W) float32."""
    assert frame_rgb.dtype == np.uint8 and frame_rgb.ndim == 3 and frame_rgb.shape[2] == 3
    t = frame_rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)      # ToTensor
    return (t - np.float32(0.5)) / np.float32(0.5)                                # Normalize


def bgr2rgb(frame: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(frame, cv2.COLOR_BGR2RGB) == cv2.COLOR_RGB2BGR: reverse the channel axis."""
    return np.ascontiguousarray(frame[..., ::-1])


def pack_inputs(frames_bgr, parsing=None) -> np.ndarray:
    """style_transfer.py:114,160,163,174: BGR frames -> `inputs = cat((x, x_p/16.), 1)`.
    frames_bgr (N,H,W,3) uint8 as VideoCapture.read delivers them; parsing (N,19,H,W) float32
    (the --parsing_map_path branch, style_transfer.py:168-169) or None."""
    x = np.stack([to_tensor_normalize(bgr2rgb(f)) for f in frames_bgr], 0)
    if parsing is None:
        return x
    return np.concatenate([x, parsing.astype(np.float32) / np.float32(16.0)], 1)


def tensor2cv2(img: np.ndarray) -> np.ndarray:
    """util.py:190-192 after torch.clamp(y_tilde, -1, 1) (style_transfer.py:177):
    (3,H,W) float32 -> (H,W,3) uint8 BGR."""
    y = np.clip(img.astype(np.float32), np.float32(-1.0), np.float32(1.0))
    tmp = ((y.transpose(1, 2, 0) + np.float32(1.0)) * np.float32(127.5)).astype(np.uint8)
    return bgr2rgb(tmp)
