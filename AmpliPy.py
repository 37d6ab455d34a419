#!/usr/bin/env python3
"""Drop-in entry point: `python AmpliPy.py <trim|variants|consensus|aio> ...` with the reference's flags,
served by the MI355X engine in amplipy_amd (see README.md).  `import AmpliPy` gives the mirror of the
reference's module-level API (run_amplipy, load_primers, load_ref_genome, find_overlapping_primers, ...)."""
from amplipy_amd.amplipy import *          # noqa: F401,F403
from amplipy_amd.amplipy import main, run_amplipy, parse_args  # noqa: F401

if __name__ == "__main__":
    main()
