"""Host-side helpers with the reference's names and behaviour (utils/helpers.py): the meta-prompt
grammar, `Rect`, the bounding-box geometry and the in-memory log.  Only what the guided-attention
path and its callers use is provided; plotting / annotation / latent statistics are out of scope.

Meta-prompt grammar (reference utils/helpers.py:59-114):
    plain words | [phrase:x,y] (COOR) | [phrase:x,y,w,h] (BOX) | [CustomLoss:name (args)] (last item)
Quirks of the reference that are kept on purpose (pinned by tests/golden/g2_parse_prompt.json):
a single trailing word after the last annotation is dropped, and plain words keep their trailing
space until the next word is appended.
"""
import os
from enum import Enum

import torch

from . import shared_state as state


class AnnotationType(Enum):
    COOR = 0
    BOX = 1
    KEYWORD = 2


class Rect:
    """Axis-aligned box; `size` is the side of the square it is expressed in (1 = fractions)."""

    def __init__(self, x, y, width, height, size):
        self.x, self.y, self.width, self.height, self.size = x, y, width, height, size

    def right(self):
        return self.x + self.width

    def bottom(self):
        return self.y + self.height

    def center(self):
        return ((self.x + self.width / 2.0), (self.y + self.height / 2.0))

    def of_size(self, new_size):
        ratio = float(new_size / self.size)
        return Rect(self.x * ratio, self.y * ratio, self.width * ratio, self.height * ratio, new_size)

    def as_tuple(self):
        return (self.x, self.y, self.width, self.height)

    def __repr__(self):
        return f"Rect({self.x}, {self.y}, {self.width}, {self.height}, size={self.size})"


def add_word(prompt, token):
    if prompt == "" or prompt.endswith(" "):
        return prompt + token
    return prompt + " " + token


def _closing_bracket(text):
    """Index in `text` of the ']' that closes text[0] == '[' (nesting aware).  Like the reference's
    findMatchingBracket (helpers.py:41-56) the scan starts two characters in, and a missing bracket
    yields 0."""
    depth = 0
    for i in range(2, len(text)):
        if text[i] == "[":
            depth += 1
        elif text[i] == "]":
            if depth == 0:
                return i
            depth -= 1
    return 0


def _annotation(text, bracket, meta_info, custom_losses):
    """Consume one `[...]` item at the head of `text`; returns (word to add or None, rest)."""
    close = _closing_bracket(text)
    colon = text.index(":")
    phrase = text[bracket + 1:colon].strip(" ")
    numbers = text[colon + 1:close].strip(" ").split(",")
    word = phrase
    if phrase == "CustomLoss":
        spec = text[colon + 1:]
        gap = spec.index(" ")
        name, args = spec[:gap], spec[gap + 1:-1]
        fn = state.config.registered_loss_functions[name]
        custom_losses[name] = (fn, args)
        for sub in fn.subprompts_of_interest(args):
            meta_info.append((sub, AnnotationType.KEYWORD, None))
        word = None
    elif len(numbers) == 2:
        meta_info.append((phrase, AnnotationType.COOR, (float(numbers[0]), float(numbers[1]))))
    elif len(numbers) == 4:
        x, y, w, h = (float(n) for n in numbers)
        meta_info.append((phrase, AnnotationType.BOX, Rect(x, y, w, h, 1)))
    return word, text[close + 1:]


def parse_prompt(meta_prompt):
    """-> (prompt, meta_info [(phrase, AnnotationType, Rect | (x, y) | None)], custom_losses)."""
    prompt, meta_info, custom_losses = "", [], {}
    text = meta_prompt
    while True:
        text = text.lstrip(" ")
        space, bracket = text.find(" "), text.find("[")
        if space < 0 and bracket < 0:
            return prompt, meta_info, custom_losses
        if bracket < 0:
            return add_word(prompt, text), meta_info, custom_losses
        if space < 0 or bracket < space:
            word, text = _annotation(text, bracket, meta_info, custom_losses)
            if word is not None:
                prompt = add_word(prompt, word)
        else:
            prompt = add_word(prompt, text[:space + 1])
            text = text[space:]


def get_meta_prompt_clean():
    s = state.config.meta_prompt
    for ch in "[]:.":
        s = s.replace(ch, "_")
    return s[0:5] if state.config.interactive else s


def get_inner_folder_name():
    return get_meta_prompt_clean()


# ---- bounding-box geometry (reference helpers.py:155-173): pixel centres, closed interval, box shrunk
sample_center = True


def inside_box(cur_x, cur_y, rect):
    if sample_center:
        cur_x += 0.5
        cur_y += 0.5
    off_x = state.curHyperParams["shrink_factor"] * rect.width
    off_y = state.curHyperParams["shrink_factor"] * rect.height
    if cur_x >= (rect.x + off_x) and cur_x <= (rect.x + rect.width - off_x):
        if cur_y >= (rect.y + off_y) and cur_y <= (rect.y + rect.height - off_y):
            return True
    return False


def inside_mask(rect, res):
    """(res, res) bool tensor of `inside_box` for a Rect already scaled with of_size(res)."""
    return torch.tensor([[inside_box(jj, ii, rect) for jj in range(res)] for ii in range(res)], dtype=torch.bool)


def get_corresponding_weight(x):
    import numpy as np
    return np.interp(x, [0, .333, .666, 1.0], [3, 2.5, 1, .2])  # hard drop off near edges (reference :159-162)


def distance_from_center(cur_x, cur_y, rect, normalized):
    import math
    if sample_center:
        cur_x += 0.5
        cur_y += 0.5
    if normalized:  # each dimension separately; 0 == at the centre, 1 == at the furthest corner
        return math.sqrt(math.pow(2 * (rect.center()[0] - cur_x) / rect.width, 2) +
                         math.pow(2 * (rect.center()[1] - cur_y) / rect.height, 2)) / math.sqrt(2)
    return math.sqrt(math.pow(rect.center()[0] - cur_x, 2) + math.pow(rect.center()[1] - cur_y, 2))


def strict_weight_table(r, res):
    """The per-pixel weights of the strict mode (reference helpers.py:216-246) for a Rect scaled with of_size(res):
    (weights (res, res) fp32 normalised separately over inside / outside, inside mask, number of inside pixels)."""
    mask = inside_mask(r, res)
    w = torch.ones(res, res)
    for ii in range(res):
        for jj in range(res):
            if mask[ii, jj]:
                w[ii, jj] = float(get_corresponding_weight(distance_from_center(jj, ii, r, True)))
    s_in = torch.zeros(())
    s_out = torch.zeros(())
    for ii in range(res):      # fp32 sums in pixel order, as the reference accumulates them
        for jj in range(res):
            if mask[ii, jj]:
                s_in = s_in + w[ii, jj]
            else:
                s_out = s_out + w[ii, jj]
    w = torch.where(mask, w / s_in, w / s_out)
    return w, mask, int(mask.sum())


def calculate_bounding_box_losses(r, imageSoftmax):
    """reference helpers.py:215-277 — (inside loss, outside loss) of the sum-normalised map for a Rect scaled to the
    map.  Non-strict: (1 - mass inside, mass outside).  Strict: the weighted hinge terms.  Stand-alone operator on any
    device; the hot path computes the same two numbers inside ga_smooth_loss_fwd."""
    res = imageSoftmax.shape[0]
    if state.curHyperParams["strict"]:
        w, mask, n_in = strict_weight_table(r, res)
        w, mask = w.to(imageSoftmax.device, imageSoftmax.dtype), mask.to(imageSoftmax.device)
        at_most = 1.0 / n_in
        zero = imageSoftmax.new_zeros(())
        inside = torch.where(mask, w * 2. * torch.clamp(at_most - imageSoftmax, min=0), zero).sum().reshape(1)
        outside = torch.where(mask, zero, w * torch.clamp(imageSoftmax, min=0)).sum().reshape(1)
        return (inside, outside)
    mask = inside_mask(r, res).to(imageSoftmax.device)
    zero = imageSoftmax.new_zeros(())
    inside = torch.where(mask, imageSoftmax, zero).sum().reshape(1)
    outside = torch.where(mask, zero, imageSoftmax).sum().reshape(1)
    return (1. - inside, outside)


def dictToString(d):
    if type(d) is dict:
        return "".join("_" + str(k) + "_" + dictToString(v) for k, v in d.items() if k != "meta_prompt")
    return str(d)


# ---- in-memory log (reference helpers.py:292-307); kept off the timed path by the pipeline's `log_level`
lines = []


def log(text, also_print=False):
    lines.append(text + os.linesep)
    if also_print:
        print(text)


def log_clear():
    global lines
    lines = []


def log_save(filename):
    with open(filename, "w") as fp:
        fp.writelines(lines)
    log_clear()


# ---- image annotation / latent statistics (reference helpers.py:129-152, 309-349): side effects of run.execute,
# off by default (config.annotate, config.diagnostic_level) and never inside the timed sampling loop
colors = ["#0000a0", "#a00000", "#00a000", "#ecf024", "#8d24f0"]


def get_color(i):
    return colors[i]


def _font(size=20):
    from PIL import ImageFont
    try:
        return ImageFont.truetype("arial.ttf", size, encoding="unic")   # what the reference asks for
    except OSError:
        try:
            return ImageFont.truetype("DejaVuSans.ttf", size)
        except OSError:
            return ImageFont.load_default()


def annotate_image(image):
    """Draw the boxes / target crosses of the meta-prompt on a PIL image when config.annotate is set (reference
    :129-152; it hard-codes 512 px and a 16-cell grid — generalised to the image's own size)."""
    if not (state.config.annotate and not state.config.interactive):
        return
    from PIL import ImageDraw
    draw = ImageDraw.Draw(image)
    font = _font()
    w, h = image.size
    for i, (word, kind, geom) in enumerate(state.config.meta_info):
        color = get_color(i % len(colors))
        if kind == AnnotationType.COOR:
            x, y, length = geom[0] * w, geom[1] * h, 15
            draw.line([(x - length, y), (x + length, y)], fill=color)
            draw.line([(x, y - length), (x, y + length)], fill=color)
            draw.text((x, y), word, fill=color, font=font)
        elif kind == AnnotationType.BOX:
            draw.rectangle([(geom.x * w, geom.y * h), (geom.right() * w, geom.bottom() * h)], fill=None, width=2,
                           outline=color)
            draw.text((w * geom.x, h * geom.y), word, fill=color, font=font)


means, stds, percentile99 = {}, {}, {}


def log_latent_stats(latent, per_channel=False):
    """mean / std of |x| / 99th percentile of |x| of the latents, appended per call (reference :313-333).  Forces a
    device->host copy: the pipeline only calls it when config.diagnostic_level > 0."""
    import numpy
    keys = [f"ch{i}" for i in range(latent.shape[1])] if per_channel else ["all"]
    for n, key in enumerate(keys):
        x = (latent[0, n] if per_channel else latent).float()
        percentile99.setdefault(key, []).append(float(numpy.quantile(x.abs().cpu().numpy(), .99)))
        stds.setdefault(key, []).append(x.abs().std().item())
        means.setdefault(key, []).append(x.mean().item())


def save_latent_stats(filename):
    """Plot the collected latent statistics (reference :335-349); only when config.diagnostic_level > 0."""
    global means, stds, percentile99
    if state.config.diagnostic_level > 0 and percentile99:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        for key in percentile99:
            plt.plot(means[key], label=f"{key} mean")
            plt.plot(percentile99[key], label=f"{key} 99")
        plt.legend(loc="best")
        plt.savefig(filename)
        plt.clf()
    means, stds, percentile99 = {}, {}, {}
