"""Golden vectors for the frame pack / unpack kernels (tests/golden/frames_io.npz).

torchvision and cv2 are absent from this image, so the three library calls the reference makes
around VToonify.forward cannot be executed; this script evaluates them with the torch ops they
are documented to be (torchvision.transforms.functional.to_tensor: `img.permute(2,0,1).to(float32)
.div(255)`; normalize: `tensor.sub_(mean).div_(std)`; cvtColor RGB<->BGR: channel reversal) and the
reference's own lines verbatim (x_p/16., torch.clamp, tensor2cv2 -- style_transfer.py:174,177,
util.py:190-192).  oracle/frames_oracle.py (numpy) must reproduce these bits.

    python tests/golden/make_golden_frames.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = torch.Generator().manual_seed(77)
    frames_bgr = torch.randint(0, 256, (2, 24, 40, 3), generator=g, dtype=torch.uint8)
    frames_bgr[0, 0, :8, :] = torch.tensor([0, 1, 2, 127, 128, 253, 254, 255], dtype=torch.uint8)[:, None]
    parsing = torch.randn(2, 19, 24, 40, generator=g) * 6.0
    xs = []
    for f in frames_bgr:
        rgb = f.flip(-1)                                                  # cv2.cvtColor(frame, COLOR_BGR2RGB)
        t = rgb.permute(2, 0, 1).contiguous().to(torch.float32).div(255)    # transforms.ToTensor
        t = t.sub_(torch.tensor([0.5, 0.5, 0.5]).view(3, 1, 1)).div_(torch.tensor([0.5, 0.5, 0.5]).view(3, 1, 1))
        xs.append(t.unsqueeze(0))
    x = torch.cat(xs, 0)
    inputs = torch.cat((x, parsing / 16.), dim=1)                           # style_transfer.py:174
    y = torch.randn(2, 3, 32, 48, generator=g) * 0.8
    y[0, :, 0, :6] = torch.tensor([-1.5, -1.0, -0.999999, 0.0, 1.0, 1.5])
    yc = torch.clamp(y, -1, 1)                                              # style_transfer.py:177
    outs = []
    for k in range(yc.size(0)):
        tmp = ((yc[k].cpu().numpy().transpose(1, 2, 0) + 1.0) * 127.5).astype(np.uint8)   # util.py:191
        outs.append(np.ascontiguousarray(tmp[..., ::-1]))                   # cv2.cvtColor(tmp, COLOR_RGB2BGR)
    np.savez_compressed(os.path.join(HERE, "frames_io.npz"), frames_bgr=frames_bgr.numpy(), parsing=parsing.numpy(),
                        inputs=inputs.numpy(), y=y.numpy(), out_bgr=np.stack(outs, 0))
    print("wrote frames_io.npz")


if __name__ == "__main__":
    main()
