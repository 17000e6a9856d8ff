timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 && \
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
