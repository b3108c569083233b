"""Visualisation helpers with the reference's names (utils/vis_utils.py).  Off the timed path: they run after a
generation, on host copies."""
import math
from typing import List

import numpy as np
import torch
from PIL import Image

from .ptp_utils import AttentionStore, aggregate_attention


def get_image_grid(images: List[Image.Image]) -> Image.Image:
    """Near-square grid, row-major (reference utils/vis_utils.py:63-73)."""
    num_images = len(images)
    cols = int(math.ceil(math.sqrt(num_images)))
    rows = int(math.ceil(num_images / cols))
    width, height = images[0].size
    grid_image = Image.new("RGB", (cols * width, rows * height))
    for i, img in enumerate(images):
        grid_image.paste(img, ((i % cols) * width, (i // cols) * height))
    return grid_image
