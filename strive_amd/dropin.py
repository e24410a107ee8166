"""Register strive_amd's mirrors under the reference's import names.

The reference's scripts run from ``src/`` and import ``models.traffic_model``, ``losses.adv_gen_nusc``,
``utils.transforms`` ... (reference src/adv_scenario_gen.py:16-31, src/refine_traffic_optim.py:19-32).

``install()`` (no arguments, what the GPU box uses: the reference's files do not exist there) makes those names resolve to
the HIP-backed modules of this package and ``torch_geometric.data`` to the minimal Data/Batch containers; importing
anything of the reference that is not on the hot path raises ImportError.

``install(reference_src='/path/to/STRIVE/src')`` is the deployment mode next to a checkout of the reference -- two lines at
the top of its driver scripts:
  * ``models.*`` and ``losses.*`` are replaced entirely (no reference code of those packages is imported);
  * ``utils`` and ``datasets`` become packages whose search path is the reference's own directory, so everything off the
    hot path (``utils.logger``, ``utils.config``, ``utils.common``, ``utils.torch``, ``datasets.nuscenes_dataset``, the
    planner ...) is the reference's unchanged file;
  * modules that mix hot-path and other functions (``utils.scenario_gen``, ``datasets.utils``, ``datasets.nuscenes_utils``)
    are the reference's files with this package's implementations laid over the same names (``datasets.map_env`` stays
    the reference's: the model only reads the raster tensors it builds);
  * the loop drivers ``utils.{init,adv_gen,sol}_optim`` are this package's by default; ``loops='reference'`` keeps the
    reference's own loop functions, which then run unchanged on top of the HIP-backed model and loss modules (that is
    what tests/test_dropin_reference.py executes in the build container).
"""
import importlib
import os
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
    'planners': 'strive_amd.planners',
    'planners.planner': 'strive_amd.planners.planner',
    'planners.hardcode_goalcond_nusc': 'strive_amd.planners.hardcode_goalcond_nusc',
}

# next to a reference checkout: fully replaced / overlaid / loop modules
_REPLACED = ('models', 'models.traffic_model', 'models.interaction_net', 'models.common', 'losses', 'losses.common',
             'losses.adv_gen_nusc', 'losses.traffic_model', 'utils.transforms')
_OVERLAID = ('datasets.utils', 'datasets.nuscenes_utils', 'utils.scenario_gen')   # in dependency order; datasets.map_env stays
# the reference's (it rasterises the maps with the devkit; the model only reads its nusc_raster / nusc_dx / bounds / L / W)
_LOOPS = ('utils.init_optim', 'utils.adv_gen_optim', 'utils.sol_optim')


def _install_pyg_stub():
    if 'torch_geometric' in sys.modules:
        return
    from . import graph
    tg = types.ModuleType('torch_geometric')
    tgd = types.ModuleType('torch_geometric.data')
    tgd.Data, tgd.Batch = graph.Data, graph.Batch
    tg.data = tgd
    sys.modules['torch_geometric'] = tg
    sys.modules['torch_geometric.data'] = tgd


def _public_names(mod):
    names = getattr(mod, '__all__', None)
    if names is None:
        names = [n for n, v in vars(mod).items()
                 if not n.startswith('_') and getattr(v, '__module__', None) == mod.__name__]
    return names


def install(with_pyg_stub=True, reference_src=None, loops='strive_amd'):
    if loops not in ('strive_amd', 'reference'):
        raise ValueError("loops must be 'strive_amd' or 'reference'")
    if reference_src is None:
        if loops == 'reference':
            raise ValueError("loops='reference' needs reference_src")
        for alias, target in _ALIASES.items():
            sys.modules[alias] = importlib.import_module(target)
        if with_pyg_stub:
            _install_pyg_stub()
        return sorted(_ALIASES)

    ref = os.path.abspath(reference_src)
    if not os.path.isdir(os.path.join(ref, 'utils')) or not os.path.isdir(os.path.join(ref, 'datasets')):
        raise FileNotFoundError('%s does not look like the reference\'s src/ directory' % ref)
    if with_pyg_stub:
        _install_pyg_stub()
    done = []
    # forget earlier registrations of the names we are about to define (e.g. the installed HF `datasets` package)
    for name in list(sys.modules):
        root = name.split('.')[0]
        if root in ('models', 'losses', 'utils', 'datasets'):
            del sys.modules[name]
    for alias in _REPLACED:
        sys.modules[alias] = importlib.import_module(_ALIASES[alias])
        done.append(alias)
    for pkg in ('utils', 'datasets'):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(ref, pkg)]          # everything we do not define is the reference's own file
        m.__package__ = pkg
        sys.modules[pkg] = m
    sys.modules['utils.transforms'] = importlib.import_module(_ALIASES['utils.transforms'])
    sys.modules['utils'].transforms = sys.modules['utils.transforms']
    for alias in _OVERLAID:
        ours = importlib.import_module(_ALIASES[alias])
        theirs = importlib.import_module(alias)        # the reference's file, found through the package path above
        for n in _public_names(ours):
            setattr(theirs, n, getattr(ours, n))
        theirs.__strive_amd_overlay__ = ours.__name__
        done.append(alias)
    for alias in _LOOPS:
        if loops == 'strive_amd':
            mod = importlib.import_module(_ALIASES[alias])
            sys.modules[alias] = mod
            setattr(sys.modules['utils'], alias.split('.')[1], mod)
        else:
            importlib.import_module(alias)
        done.append(alias)
    return sorted(done)
