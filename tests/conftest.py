import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference built in the development container)")


def pytest_collection_modifyitems(config, items):
    import oracle_lib
    if not oracle_lib.have_ref():
        skip = pytest.mark.skip(reason="oracle/_ref not built (reference sources only exist in the dev container)")
        for it in items:
            if "ref" in it.keywords:
                it.add_marker(skip)
