"""Name -> class lookup with the reference's rule (reference src/models/ModelFactory.py:10-22): module
`models.<name>`, class `<name[:-2]>`.  Present so this tree is usable on its own; inside the reference tree the
reference's own factory finds VipNeRFHip01.py the same way."""
import importlib
import inspect


def get_model(configs: dict, model_configs: dict = None):
    filename = configs['model']['name']
    classname = filename[:-2]
    module = importlib.import_module(f'models.{filename}')
    for name, cls in inspect.getmembers(module, inspect.isclass):
        if name == classname:
            return cls(configs, model_configs)
    raise RuntimeError(f'Unknown model: {filename}')
