"""Process-global state of one in-flight generation — the same names the reference keeps in
utils/shared_state.py (its loss code and pipeline read them as module attributes).  One generation
per process at a time; the seed-parallel driver runs one process per GPU."""

config = None                # the RunConfig of the current run (token_dict, prompt, ... attached by run.parseMetaPrompt)
cur_seed = None
cur_time_step_iter = None    # denoising step index i
always_save_iter = [24, 25, 26]
sub_iteration = 0            # refinement sub-iteration at the current step

sigmas = None                # sqrt((1 - a)/a) per training timestep
timesteps = None             # the 50 inference timesteps 981, 961, ..., 1

# deep-feature optimisation is experimental and off in the reference (shared_state.py:10-15); the flags
# exist so that code probing them keeps working, the branch itself is out of scope
optimizeDeepLatent = False
use_loss_total = True

curHyperParams = None

# reference utils/shared_state.py:21-22 — note that `thresholds` here overrides RunConfig.thresholds
hyperParameterOverrides = {"strict": False, "inside_loss_scale": .2, "outside_loss_scale": .2, "shrink_factor": .15,
                           "thresholds": {0: 1.}, "use_optimizer": False, "recurse_until": 14, "recurse_steps": 3}
hyperParameterIterations = [{}]


def get_sigma():
    return sigmas[timesteps[cur_time_step_iter]]


def get_hyperparam_states():
    """One merged dict per entry of hyperParameterIterations (reference :29-36)."""
    return [{**hyperParameterOverrides, **over} for over in hyperParameterIterations]


tags = ["cur_seed", "cur_time_step_iter", "optimizeDeepLatent"]


def to_str(t):
    return "{:02d}".format(t) if type(t) is int else str(t)


def get_name():
    return "".join(f"{t}_{to_str(globals()[t])}_" for t in tags)
