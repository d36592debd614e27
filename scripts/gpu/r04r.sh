python -c "import __graft_entry__ as g; g.smoke()"
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2; done
