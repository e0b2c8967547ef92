#!/bin/bash
# One command for whoever has a network and docker: builds the reference's own scanMatching against the real Eigen 3.3 /
# Ceres 2.0 (shared) / Open3D 0.12 / yaml-cpp / ROS melodic (oracle/ref_harness/Dockerfile), runs it on the inputs of the
# committed golden cases and writes tests/golden_ref/case_*.ref.txt.  Afterwards:
#     python -m pytest tests/test_golden_ref.py            # the C oracle and the golden vectors against the reference
#     python -m pytest tests/test_golden_ref.py -m gpu     # ... and the HIP path (on an MI355X)
# TEST INFRASTRUCTURE; never run in the round's image (no network, no docker).
set -euo pipefail
cd "$(dirname "$0")/../.."
python tests/golden_ref_tools/export_ref_inputs.py oracle/_ref/in     # tests/golden/case_*.npz -> the harness's flat inputs
docker build -f oracle/ref_harness/Dockerfile -t tloam-ref-pin ${TLOAM_REF:+--build-arg TLOAM_REF=$TLOAM_REF} .
docker run --rm -v "$PWD":/work tloam-ref-pin
echo "tests/golden_ref/: $(ls tests/golden_ref | wc -l) files; reference commit $(cat tests/golden_ref/REFERENCE_COMMIT)"
