"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

A CPU restatement (plain torch fp32 / float64 index math) of the algorithm on STRIVE's
latent-optimisation hot path: the CVAE traffic prior's embed + autoregressive decoder rollout, the
rasterised-map lookups, and the collision / off-road / prior losses with their Adam loops.  Every
function cites the reference file:line it restates.

Who may import this package: tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker / reported baseline only.  ``strive_amd`` (the product) never imports
it and fails loudly when its HIP library is missing.

How it is pinned: the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself, produced in the
build container by ``tests/golden/make_golden.py`` (which imports /root/reference/src with import
stand-ins for the absent third-party wheels) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks the oracle against every one of them.  The one boundary that
stays *unpinned by the upstream project* is torch-geometric 1.7.1 / torch-scatter 2.0.7's
gather + scatter-max (wheels absent, no reference test covers it): the stand-in used when
generating the vectors restates its published semantics -- see ``oracle/model.py``.
"""
