"""TEST INFRASTRUCTURE ONLY — CPU oracle for the CnnOcrModel CNN->BiLSTM->CTC hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.  The product path
(``vistaocr_amd``) never imports this package and fails loudly when its HIP library is missing.
"""
