"""CPU oracle for the ConsistentID UNet-denoise hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 restatement of the reference's algorithm for
the hot path named in BASELINE.json.  It exists so that the
hand-written HIP path in ``consistentid_amd/`` can be checked against something.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
    leg may import anything from here;
  * nothing under ``consistentid_amd/`` may import, call or fall back to it.

PARITY PIN STATUS
  * ``oracle/processors.py`` (the reference's own arithmetic, attention.py:90-294)
    is PINNED: ``tests/golden/make_golden.py`` imports the real
    ``/root/reference/attention.py`` (with import shims for the three third-party
    symbols it pulls in) and stores input/output vectors under ``tests/golden/``;
    ``tests/test_oracle_golden.py`` replays them through this restatement.
  * ``oracle/unet.py``, ``oracle/ddim.py``, ``oracle/loop.py`` restate
    diffusers==0.23.0 (reference requirements.txt:36), which is NOT vendored under
    /root/reference and not installed in the build container: for those files
    "parity unpinned" -- anchored only on the reference's call sites and on
    structural invariants (parameter counts, processor enumeration, closed forms).
"""
