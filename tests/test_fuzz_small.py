"""A few seeds of the differential fuzzers of tools/ in the test suite: the CLI against the unmodified reference binary (oracle/_ref/cornetto, built by
`oracle/ref.mk`; it travels with the snapshot) — the host path here (CORNETTO_ACCEL=no), the device path on the GPU box — and the coverage stage and
the drop-in sdust entry points against the oracle / the reference's sdust().  The long campaigns are `python tools/fuzz_*.py` (profiles/README.md)."""
import os
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = os.path.join(ROOT, "oracle", "_ref", "cornetto")


def _cli_fuzz(kind, seeds, monkeypatch, host):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/cornetto not built")
    import fuzz_cli
    if host:
        monkeypatch.setenv("CORNETTO_ACCEL", "no")
    fn = {"fasta": fuzz_cli.fuzz_fasta, "telo": fuzz_cli.fuzz_telo, "panel": fuzz_cli.fuzz_panel, "bigenough": fuzz_cli.fuzz_bigenough}[kind]
    with tempfile.TemporaryDirectory() as tmp:
        for seed in seeds:
            ok, info = fn(seed, tmp)
            assert ok, (kind, seed, info)


@pytest.mark.parametrize("kind", ["fasta", "telo", "panel", "bigenough"])
def test_cli_host_path_against_the_reference_binary(kind, monkeypatch):
    _cli_fuzz(kind, range(910_000, 910_012), monkeypatch, host=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fasta", "telo", "panel"])
def test_cli_device_path_against_the_reference_binary(kind, monkeypatch):
    # (920000 and 80066 of the panel cases: a negative depth value — on one device, and with the contigs dealt to several handles)
    _cli_fuzz(kind, list(range(920_000, 920_016)) + ([80_066] if kind == "panel" else []), monkeypatch, host=False)


@pytest.mark.gpu
def test_coverage_stage_and_sdust_entry_points_on_a_few_seeds(monkeypatch):
    import fuzz_abi_sdust
    import fuzz_cov
    monkeypatch.setattr(sys, "argv", ["fuzz_cov.py", "930000", "40"])
    assert fuzz_cov.main() == 0
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libcornetto_ref.so")):
        monkeypatch.setattr(sys, "argv", ["fuzz_abi_sdust.py", "940000", "40"])
        assert fuzz_abi_sdust.main() == 0
