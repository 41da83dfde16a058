"""``ldm.lr_scheduler.LambdaLinearScheduler`` is named by configs/train.yaml:23 but only stored by the model at
inference (training-only); kept importable so the unchanged YAML loads."""


class LambdaLinearScheduler:
    def __init__(self, warm_up_steps=None, f_min=None, f_max=None, f_start=None, cycle_lengths=None, verbosity_interval=0):
        self.warm_up_steps, self.f_min, self.f_max, self.f_start, self.cycle_lengths = warm_up_steps, f_min, f_max, f_start, cycle_lengths

    def __call__(self, n, **kwargs):
        raise NotImplementedError("learning-rate schedules are training-only (out of scope)")
