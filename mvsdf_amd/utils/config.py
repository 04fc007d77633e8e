"""Dict-backed stand-in for pyhocon's ConfigTree (pyhocon is not in the image) + a reader for the HOCON subset used by
reference code/confs/mvsdf_dtu.conf (nested blocks `name { ... }` / `name\\n{`, `key = value`, lists, numbers, bools, fractions
kept as strings).  IDRNetwork only needs get_int / get_float / get_config (idr.py:172-177)."""
import re


class ConfigDict(dict):
    def _get(self, key):
        node = self
        for part in key.split('.'):
            node = node[part]
        return node

    def get_int(self, key): return int(self._get(key))
    def get_float(self, key): return float(self._get(key))
    def get_string(self, key): return str(self._get(key))
    def get_list(self, key): return list(self._get(key))
    def get_bool(self, key): return bool(self._get(key))

    def get_config(self, key):
        v = self._get(key)
        return v if isinstance(v, ConfigDict) else ConfigDict(v)


def _value(tok):
    tok = tok.strip()
    if tok.startswith('['):
        inner = tok[1:-1].strip()
        return [_value(t) for t in inner.split(',')] if inner else []
    if tok in ('True', 'true'):
        return True
    if tok in ('False', 'false'):
        return False
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok.strip('"')


def parse_hocon(text):
    """Parse the HOCON subset of mvsdf_dtu.conf into nested ConfigDicts."""
    text = re.sub(r'#.*|//.*', '', text)
    tokens = re.findall(r'\[[^\]]*\]|[{}]|[^\s{}=]+\s*=\s*\[[^\]]*\]|[^\s{}=]+\s*=\s*[^\s{}]+|[^\s{}=]+', text)
    root = ConfigDict()
    stack, pending = [root], None
    for tok in tokens:
        if tok == '{':
            child = ConfigDict()
            stack[-1][pending] = child
            stack.append(child)
            pending = None
        elif tok == '}':
            stack.pop()
        elif '=' in tok:
            k, v = tok.split('=', 1)
            stack[-1][k.strip()] = _value(v)
        else:
            pending = tok
    return root


def load_conf(path):
    with open(path) as f:
        return parse_hocon(f.read())
