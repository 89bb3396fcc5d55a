"""CPU oracle for the cine-MRI reconstruction hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (PyTorch fp32 / numpy)
restatement of the reference algorithm (f78bono/deep-cine-cardiac-mri,
``reconstruction/{models,utils}``).  Every function cites the reference
file:line it follows.  It exists so that the hand-written HIP path can be
checked against something that runs anywhere.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU
baseline, never as the thing shipped.  Nothing under
``deep-cine-cardiac-mri_amd/`` imports it; the product path has no CPU
fallback and raises when the HIP library is missing.

Parity pin: the reference ships no tests or golden vectors of its own
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, imported in the build container from /root/reference by
``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks the oracle
against every one of them.
"""

from . import centered_fft, complex_ops  # noqa: F401
