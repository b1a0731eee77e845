import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def known_answers():
    import json
    with open(os.path.join(GOLDEN, "known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture(autouse=True)
def _oracle_reading_of_the_edit_distance_back_to_default():
    """a test may switch the oracle's grouping to the second reading of the absent edit distance (orc.set_edit_free_end):
    whatever happens in it, the next test starts from the default"""
    yield
    orc = sys.modules.get("oracle.orc")
    if orc is not None and getattr(orc, "_LIB", None) is not None:
        orc.set_edit_free_end(0)
