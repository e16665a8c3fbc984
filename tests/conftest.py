import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "changes_env: the test sets process environment variables (runs before tests/prefetch.py's threads start)")


def _prefetching():
    """JTK_PREFETCH=1 turns tests/prefetch.py on (off by default: see its header)."""
    return os.environ.get("JTK_PREFETCH") == "1" and not os.environ.get("JTK_DEVICE_IS_ORACLE")


@pytest.hookimpl(trylast=True)   # (after -m / -k have deselected)
def pytest_collection_modifyitems(config, items):
    if not _prefetching():
        return   # the suite in its plain order, one device call at a time
    # bench.py's multi-rank tests first, while this process has not touched the GPU (eight child processes beside a parent that
    # holds device memory and hardware queues did not finish in 20 minutes); then the tests that setenv (the threads of
    # tests/prefetch.py, whose library calls getenv, start after the last of them); then the rest
    def rank(it):
        if os.path.basename(str(it.fspath)) == "test_bench_gpu.py":
            return 0
        return 1 if it.get_closest_marker("changes_env") else 2
    items[:] = sorted(items, key=rank)   # (stable: the order inside each group stays)
    config._jtk_selected = [it.nodeid for it in items]


def pytest_runtest_setup(item):
    """GPU runs only: once the environment-changing tests are through, the suite's slowest device calls start on other host
    threads (tests/prefetch.py) and the tests that need them pick the results up."""
    if item.get_closest_marker("gpu") is None or item.get_closest_marker("changes_env") is not None:
        return
    if os.path.basename(str(item.fspath)) == "test_bench_gpu.py":
        return   # bench.py's child processes (up to eight ranks on this one GPU) get the device to themselves
    if not _prefetching():
        return
    import prefetch
    if prefetch._pool is not None:
        return
    from jtk_amd import ffi
    try:
        if ffi.lib().jtk_lc_device_ok(0) != 1:
            return
    except Exception:
        return
    prefetch.install()
    prefetch.start(getattr(item.config, "_jtk_selected", None))


def pytest_sessionfinish(session, exitstatus):
    if "prefetch" in sys.modules:
        sys.modules["prefetch"].stop()


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def jtk_lib():
    """libjtk_lc.so; built in-tree with hipcc if missing (cross-compiles without a GPU)."""
    from jtk_amd import build, ffi
    build.build()
    return ffi.lib()


@pytest.fixture(scope="session", autouse=True)
def _oracle_fixtures():
    """The oracle's answers for the large cases come from tests/golden/oracle/ (helpers.cached_cluster_chunks).  With
    JTK_DEVICE_IS_ORACLE=1 (tests/golden/make_oracle_cache.py) the oracle also stands in for the device, so that the fixtures can be
    regenerated where there is no GPU: the device calls of the listed tests return the oracle's own answer."""
    import helpers
    import oracle_ffi
    if not getattr(oracle_ffi.cluster_chunks, "_cached", False):
        oracle_ffi.cluster_chunks = helpers.cached_cluster_chunks(oracle_ffi.cluster_chunks)
        oracle_ffi.cluster_chunks._cached = True
    if os.environ.get("JTK_DEVICE_IS_ORACLE"):
        from jtk_amd import api, ffi
        # a generator switch for machines WITHOUT a GPU: on a box that has one it would turn every large parity test into
        # oracle-vs-oracle and report green, so it is refused there
        try:
            has_device = ffi.lib().jtk_lc_device_ok(0) == 1
        except Exception:
            has_device = False
        if has_device:
            pytest.exit("JTK_DEVICE_IS_ORACLE is set on a machine with a gfx950 device: the GPU tests would compare the oracle with "
                        "itself.  Unset it (it exists for tests/golden/make_oracle_cache.py on CPU-only machines).", returncode=3)

        def dev_chunks(params, batch, device=0, raise_on_chunk_failure=True, devices=None):
            out = dict(oracle_ffi.cluster_chunks(helpers.oracle_params(params), batch, skip_polish=False))
            out.setdefault("rc", 0)
            return out

        def dev_features(params, feature_chunks, variants, variant_type, post_stride, device=0, **kw):
            n = int(feature_chunks["n_reads"].sum())
            return helpers.oracle_cluster_features(helpers.oracle_params(params), feature_chunks, variants, variant_type, post_stride, n)
        api.cluster_chunks = dev_chunks
        api.cluster_features = dev_features
    yield
