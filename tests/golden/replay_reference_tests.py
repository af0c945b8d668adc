#!/usr/bin/env python
"""Run the REFERENCE's own host-side unit tests against ``tgm_amd`` (build container only: the test files are read in place from
/root/reference/test/unit and never copied; nothing here travels to the GPU box).

``import tgm`` -- and every ``tgm.*`` path the test files use -- resolves to ``tgm_amd``: each ``tgm_amd`` module is registered in
``sys.modules`` under the reference's name before the test files are imported, so what runs is the reference's assertions over
our DGData (validation, sort, discretize, splits) / DGraph / DGBatch / DGDataLoader / HookManager / registry / DeduplicationHook.  Cases the reference marks ``gpu``
are excluded (no device here; tests/test_dgraph_views.py covers the device store against fixture g13).  The files are the
host-side surface of SURVEY.md section 8 rows a1-a7 / a16 -- the sampler / aggregation tests of the reference pass CPU tensors,
which ``tgm_amd`` refuses by design (no CPU fallback); those rows are covered by the g1-g12 fixtures on the device.

    python tests/golden/replay_reference_tests.py        # prints "<passed> / <total>", exit code 1 unless all pass
"""
from __future__ import annotations

import importlib
import os
import pkgutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_TESTS = '/root/reference/test/unit'
FILES = [
    'test_data/test_dataloader.py',
    'test_hooks/test_hook_manager.py',
    'test_hooks/test_registry.py',
    'test_core/test_dgraph.py',
    'test_hooks/test_deduplication_hook.py',
    'test_util/test_seed.py',
    'test_data/test_data.py',
    'test_data/test_split.py',
    'test_core/test_timedelta.py',
]
# Ingest from CSV / pandas / the TGB packages is out of scope (host-side, one-shot; the TGB packages are not in the image):
# the cases of test_data.py / test_split.py that go through it (and test_timedelta.py's two checks of the per-data-set unit tables) are deselected BY NAME -- everything else in the two files (DGData validation,
# normalisation, sort, node / edge types, discretize, clone, the split strategies) runs.
DESELECT = 'not from_csv and not from_pandas and not tgb and not thgl and not tkgl'


class _Tally:
    def __init__(self) -> None:
        self.passed, self.failed = 0, []

    def pytest_runtest_logreport(self, report) -> None:
        if report.when == 'call' and report.passed:
            self.passed += 1
        elif report.failed:
            self.failed.append(report.nodeid)

    def pytest_collectreport(self, report) -> None:
        if report.failed:
            self.failed.append(f'collection: {report.nodeid}')


def alias_package() -> None:
    sys.dont_write_bytecode = True
    sys.path.insert(0, REPO)
    import tgm_amd

    sys.modules['tgm'] = tgm_amd
    for m in pkgutil.walk_packages(tgm_amd.__path__, 'tgm_amd.'):
        mod = importlib.import_module(m.name)
        sys.modules['tgm' + m.name[len('tgm_amd'):]] = mod
    # names registered by hand in sys.modules (the reference's import paths, tgm_amd/nn/encoder/__init__.py) follow too
    for name, mod in list(sys.modules.items()):
        if name.startswith('tgm_amd.'):
            sys.modules.setdefault('tgm' + name[len('tgm_amd'):], mod)


def main() -> int:
    import pytest

    alias_package()
    tally = _Tally()
    args = ['-p', 'no:cacheprovider', '-o', 'addopts=', '-c', os.devnull, '--rootdir', REF_TESTS, '-q', '-m', 'not gpu', '-k', DESELECT, '-W', 'ignore',
            *(os.path.join(REF_TESTS, f) for f in FILES)]  # fmt: skip
    rc = pytest.main(args, plugins=[tally])
    total = tally.passed + len(tally.failed)
    print(f'[replay] reference unit tests against tgm_amd: {tally.passed} / {total} ({", ".join(FILES)})', flush=True)
    for f in tally.failed:
        print(f'[replay] FAILED {f}', flush=True)
    return 0 if (rc == 0 and not tally.failed and tally.passed > 0) else 1


if __name__ == '__main__':
    sys.exit(main())
