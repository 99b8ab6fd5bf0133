export PYTHONPATH=.
timeout 600 python -m pytest tests/test_half_final.py -m gpu -x -q 2>&1 | grep -E "^E|Error|assert" | head -12
python - <<'PY'
import pire_amd
from tests import helpers as H
for c in H.golden()["half_final"]:
    t = pire_amd.Table(H.load_blob(c["blob"]))
    i = t.info
    print(c["name"], c["pattern"], "states", i.states, "letters", i.letters, "regexps", i.regexps)
PY
