"""Evaluation-time image transform of the v2/v3/v4 detectors with the names of the reference's
data/transformers.py (:456-471).  The reference resizes and normalises one image at a time on the host with
cv2 / numpy; here a whole list of images becomes the network's input tensor in one kernel launch."""
import numpy as np
import torch

from ..preprocess import preprocess_u8


def base_transform(image, size, mean, std):
    """uint8 HWC image -> float32 HWC tensor on the GPU: resize to size x size, / 255, - mean, / std, channel order
    unchanged (reference :456-461)."""
    x = preprocess_u8([image], [(size, size, 0, 0)], (size, size), swap_rb=False, scale_mode=2, mean=mean, std=std,
                      channels_last=True)
    return x[0]


class BaseTransform:
    def __init__(self, size, mean=(0.406, 0.456, 0.485), std=(0.225, 0.224, 0.229)):
        self.size = size
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)

    def __call__(self, image, boxes=None, labels=None):
        return base_transform(image, self.size, self.mean, self.std), boxes, labels

    def batch(self, images, dtype=torch.float32, device=None):
        """List of BGR uint8 HWC images -> [N, 3, size, size] RGB, normalised: the transform followed by the dataset's
        img[:, :, (2, 1, 0)] and permute(2, 0, 1) (reference data/dataset.py:224-229), for the whole list at once."""
        s = self.size
        return preprocess_u8(images, [(s, s, 0, 0)] * len(images), (s, s), device=device, swap_rb=True, scale_mode=2,
                             mean=self.mean, std=self.std, dtype=dtype)
