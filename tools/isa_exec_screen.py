"""Command-line front of recguru_amd/isa_screen.py (the static ISA screens every build runs):

    python tools/isa_exec_screen.py recguru_amd/build/isa/*.s      the ISA the last build kept
    python tools/isa_exec_screen.py --build                        recompile every kernel file to ISA first

Exit status 1 if any kernel is flagged (a join block that runs vector instructions in front of its exec restore; a packed-f32
operation that feeds a low result from the high half of its second source)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd.isa_screen import main

if __name__ == "__main__":
    main()
