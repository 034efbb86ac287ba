"""CPU oracle for the Performer RNA-encoder path.

TEST INFRASTRUCTURE ONLY. Nothing under ``scmomer_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.
"""
