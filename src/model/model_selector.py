# Same entry points as the reference's src/model/model_selector.py (model_selector / optimizer_selector / scheduler_selector).
from pathlib import Path
from runpy import run_path

import torch

from dualpixelface_amd.selectors import optimizer_selector, scheduler_selector  # noqa: F401


def model_selector(option):
    loaded = run_path(str(Path('src/model') / option.model_name / 'mainmodel.py'))
    model = loaded[option.model_name.upper()](option)
    if option.load_model is not None and option.mode != 'train':
        ckpt = torch.load(option.load_model, map_location='cpu')
        if 'state_dict' in ckpt:
            weights = ckpt['state_dict']
        elif 'model' in ckpt:
            weights = ckpt['model']
        else:
            raise NotImplementedError('wrong checkpoint')
        model.load_state_dict(weights, strict=option.load_strict)
    return model
