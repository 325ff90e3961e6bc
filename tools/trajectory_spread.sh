#!/bin/bash
# Plateau of the census term (steps 60-120) of tests/test_hip_train.py::test_training_learns_a_known_motion_like_the_reference, three cases per
# run, six runs per setting of the fused-loss switches: is a change of the loss side visible in the spread of the (chaotic) trajectories?
for cfg in "${@:-X=1}"; do echo "== $cfg"; for i in 1 2 3 4 5 6; do env $cfg python -m pytest tests/test_hip_train.py -x -q -s -m gpu -k "learns_a_known_motion" 2>&1 | grep -E "census_loss: plateau|failed" | sed "s/census_loss: plateau (steps 60-120)//; s/, reference 0.3403//" | tr "\n" " "; echo; done; done
