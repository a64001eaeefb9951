"""The reference's layered JSON configuration (config_/config_manager.py:53-95) without its side effects (no workspace
directories, no logger): config_/<config>.json + src/model/<model>/<model_config>.json under ``model`` +
dataloader/<dataset>/<dataset_config>.json under ``dataset`` + the augmentation blocks named by ``augmentation`` from
dataloader/preprocess/<augmentation_config>.json -> recursive attribute object."""
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Option(object):
    def __init__(self, d):
        for key, value in d.items():
            if isinstance(value, (list, tuple)):
                setattr(self, key, [Option(x) if isinstance(x, dict) else x for x in value])
            else:
                setattr(self, key, Option(value) if isinstance(value, dict) else value)


def load_option(config='train_faceDP', root=None, **model_overrides):
    root = root or _ROOT
    data = {'load_model': None}
    data.update(json.load(open(os.path.join(root, 'config_', config + '.json'))))
    data['sync_batch'] = data.get('accelerator') == 'ddp'
    data['model'] = json.load(open(os.path.join(root, 'src', 'model', data['model_name'], data['model_config'] + '.json')))
    data['dataset'] = json.load(open(os.path.join(root, 'dataloader', data['dataset_name'], data['dataset_config'] + '.json')))
    if 'augmentation' in data:                                   # config_manager.py:80-85: copy the named augmentation blocks
        preprocess = json.load(open(os.path.join(root, 'dataloader', 'preprocess', data['augmentation_config'] + '.json')))
        for aug in data['augmentation']:
            if aug in preprocess:
                data[aug] = preprocess[aug]
    data['model'].update(model_overrides)
    return Option(data)
