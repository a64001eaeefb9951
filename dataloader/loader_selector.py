"""Same entry point as the reference's dataloader/loader_selector.py:7-17: `loader_selector(option, training)` returns the dataset
object `<dataset_name>Loader(option, training)` defined in dataloader/<dataset_name>/loader.py."""
from pathlib import Path
from runpy import run_path

_HERE = Path(__file__).resolve().parent


def loader_selector(option, training):
    path = _HERE / option.dataset_name / 'loader.py'
    if not path.is_file():
        raise NotImplementedError('dataloader selector : %s is not implemented' % option.dataset_name)
    return run_path(str(path))[option.dataset_name + 'Loader'](option, training)
