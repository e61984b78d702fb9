#!/bin/bash
# call 15: soak, part 3 -- the FULL GPU suite 22 x on the round's final tree (kernel sources f89440a0dbb8f1c7)
bash tools/calls/r05/soak.sh full_suite_final 22 X=1 -- tests -m gpu
