"""Register strive_amd's mirrors under the reference's import names.

The reference's scripts run from ``src/`` and import ``models.traffic_model``, ``losses.adv_gen_nusc``,
``utils.transforms`` ... (reference src/adv_scenario_gen.py:16-31, src/refine_traffic_optim.py:19-32).  After
``strive_amd.dropin.install()`` those names resolve to the HIP-backed modules of this package, and
``torch_geometric.data`` resolves to the minimal Data/Batch containers (PyG is only used for collation on this path).
Only names on the hot path are provided; importing anything else of the reference raises ImportError as before.
"""
import importlib
import sys
import types

_ALIASES = {
    'models': 'strive_amd.models',
    'models.traffic_model': 'strive_amd.models.traffic_model',
    'models.interaction_net': 'strive_amd.models.interaction_net',
    'models.common': 'strive_amd.models.common',
    'losses': 'strive_amd.losses',
    'losses.common': 'strive_amd.losses.common',
    'losses.adv_gen_nusc': 'strive_amd.losses.adv_gen_nusc',
    'losses.traffic_model': 'strive_amd.losses.traffic_model',
    'utils': 'strive_amd.utils',
    'utils.transforms': 'strive_amd.utils.transforms',
    'utils.scenario_gen': 'strive_amd.utils.scenario_gen',
    'utils.init_optim': 'strive_amd.utils.init_optim',
    'utils.adv_gen_optim': 'strive_amd.utils.adv_gen_optim',
    'utils.sol_optim': 'strive_amd.utils.sol_optim',
    'datasets': 'strive_amd.datasets',
    'datasets.utils': 'strive_amd.datasets.utils',
    'datasets.nuscenes_utils': 'strive_amd.datasets.nuscenes_utils',
    'datasets.map_env': 'strive_amd.datasets.map_env',
}


def install(with_pyg_stub=True):
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
    if with_pyg_stub and 'torch_geometric' not in sys.modules:
        from . import graph
        tg = types.ModuleType('torch_geometric')
        tgd = types.ModuleType('torch_geometric.data')
        tgd.Data, tgd.Batch = graph.Data, graph.Batch
        tg.data = tgd
        sys.modules['torch_geometric'] = tg
        sys.modules['torch_geometric.data'] = tgd
    return sorted(_ALIASES)
