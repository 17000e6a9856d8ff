timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 && bash tools/ablate.sh
