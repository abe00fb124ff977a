"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32 / numpy) restatement of the reference's sampling hot path, used as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under kandinsky-2_amd/ imports
it; the product path has no CPU fallback.

Parity pinning: the reference ships no tests or golden vectors, so the restatement is
pinned against the reference's OWN modules, imported from /root/reference in the build container by
oracle/ref_loader.py (oracle/make_golden.py writes tests/golden/*.pt from those runs, and
tests/test_host_cpu.py re-checks the restatement against the committed fixtures everywhere).
"""
