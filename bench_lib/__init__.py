"""bench.py's parts: job (the prover job and its step functions), legs (what the default line carries beside `value`),
multi (N > 1), report (per-kernel table, roofline, box fingerprint).  bench.py at the repo root is the entry point and
re-exports the names tools and tests use."""
