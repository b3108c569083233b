"""`RunConfig`: the flag set of the reference (config.py:6-58), same names and defaults."""
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List


@dataclass
class RunConfig:
    meta_prompt: str                       # e.g. 'a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]'
    sd_2_1: bool = False                   # use the SD-2.1-base shapes / EOT-normalised text slice
    seeds: List[int] = field(default_factory=lambda: [42])
    output_path: Path = Path("./outputs")
    n_inference_steps: int = 50
    guidance_scale: float = 7.5
    max_iter_to_alter: int = 25            # denoising steps that may apply guided attention
    attention_res: int = 16                # resolution of the attention maps the loss reads
    run_standard_sd: bool = False
    # step -> threshold for iterative refinement; NOTE shared_state.hyperParameterOverrides["thresholds"]
    # replaces this at run time exactly as in the reference (run.py:75-79)
    thresholds: Dict[int, float] = field(default_factory=lambda: {0: 0.1, 3: 0.8})
    scale_factor: int = 20
    scale_range: tuple = field(default_factory=lambda: (1.0, 0.5))
    smooth_attentions: bool = True
    sigma: float = 0.5
    kernel_size: int = 3
    save_cross_attention_maps: bool = False
    half_precision: bool = False
    interactive: bool = False
    diagnostic_level: int = 0
    annotate: bool = False
    sub_prompt_avg_within: bool = False
    save_all_maps: bool = False
    save_individual_CA_maps: bool = False
    only_update_on_threshold_steps: bool = True

    def __post_init__(self):
        self.output_path = Path(self.output_path)
        self.output_path.mkdir(exist_ok=True, parents=True)
