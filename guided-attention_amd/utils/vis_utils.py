"""Visualisation helpers with the reference's names (utils/vis_utils.py).  Off the timed path: they run after a
generation, on host copies."""
import math
from typing import List

import numpy as np
import torch
from PIL import Image

from .ptp_utils import AttentionStore, aggregate_attention


def get_image_grid(images: List[Image.Image]) -> Image.Image:
    """The images of a run on one sheet, filled row by row; the sheet is as square as the count allows (ceil(sqrt(n))
    columns), unused cells stay black — what run.py:131 saves next to the per-seed images (reference :63-73)."""
    cols = math.isqrt(max(len(images) - 1, 0)) + 1
    rows = -(-len(images) // cols)
    cell_w, cell_h = images[0].size
    sheet = Image.new("RGB", (cols * cell_w, rows * cell_h))
    for k, im in enumerate(images):
        r, c = divmod(k, cols)
        sheet.paste(im, (c * cell_w, r * cell_h))
    return sheet


def show_image_relevance(image_relevance, image: Image.Image, relevnace_res=16):
    """Heat-map of one token's map over the image (reference :38-60: bilinear up-sampling to res^2 pixels, min-max
    normalisation, JET colours added to the normalised image).  The colour map comes from matplotlib (OpenCV is not
    required); returned in the reference's channel order (it converts RGB -> BGR at the end)."""
    from matplotlib import cm
    size = relevnace_res ** 2
    image = np.array(image.resize((size, size)))
    rel = image_relevance.reshape(1, 1, image_relevance.shape[-1], image_relevance.shape[-1]).float()
    rel = torch.nn.functional.interpolate(rel, size=size, mode="bilinear").cpu()
    rel = (rel - rel.min()) / (rel.max() - rel.min())
    rel = rel.reshape(size, size).numpy()
    image = (image - image.min()) / (image.max() - image.min())
    heat = cm.jet(np.uint8(255 * rel))[..., :3][..., ::-1].astype(np.float32)   # BGR like cv2.applyColorMap
    cam = heat + np.float32(image)
    cam = cam / np.max(cam)
    return np.uint8(255 * cam)[..., ::-1]


def show_cross_attention(prompt: str, attention_store: AttentionStore, tokenizer, indices_to_alter: List[int], res: int,
                         from_where: List[str], select: int = 0, orig_image=None, display_image=True):
    """One heat-map panel per token to alter, captioned with the token (reference :12-35).  Reads the aggregated maps
    back to the host: a diagnostic, never called inside the sampling loop.  Returns the PIL grid."""
    from . import ptp_utils
    tokens = tokenizer(prompt)["input_ids"]
    attention_maps = aggregate_attention(attention_store, res, from_where, True, select).detach().cpu()
    images = []
    for i in range(len(tokens)):
        if i in indices_to_alter:
            image = show_image_relevance(attention_maps[:, :, i], orig_image)
            image = np.array(Image.fromarray(image.astype(np.uint8)).resize((res ** 2, res ** 2)))
            images.append(ptp_utils.text_under_image(image, tokenizer.decode(int(tokens[i]))))
    return ptp_utils.view_images(np.stack(images, axis=0), display_image=display_image)
