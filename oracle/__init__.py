"""CPU oracle for the ISubGVQA inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this
directory.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

The oracle is a plain PyTorch-CPU (fp32) restatement of the reference's
algorithm for the path named in BASELINE.json.  Every function cites the
reference file:line it follows.  The sparse arithmetic of the reference lives
in third-party packages that are absent from /root/reference
(torch_geometric==2.6.1, torch_scatter==2.1.2, requirements.txt:14-15); their
published semantics are restated in ``primitives.py``.

Parity pin (see DESIGN.md "Oracle"):
  * samplers, question encoder/decoder: pinned against outputs of the real
    reference modules imported from /root/reference (tests/golden/g1..g4).
  * MGAT / conv / masking / pooling glue: pinned against the real reference
    glue code run over ``pyg_standin.py`` (our restatement of the six absent
    third-party primitives) -> tests/golden/g5*.  The primitives themselves are
    pinned by hand-computed known answers and a dense per-graph brute force.
"""
