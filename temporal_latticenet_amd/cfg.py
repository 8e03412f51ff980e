"""Minimal hjson reader for the reference's `.cfg` files (seq_config/*.cfg) and a drop-in for the
reference's cfgParser.py:3-74 (same getter names).  hjson itself is not a dependency: the subset used
by the reference is `key: value` members, nested `{}` / `[]`, quoted strings, numbers, true/false/null,
`//`, `#` and `/* */` comments, optional commas and an optional root brace."""
import re

__all__ = ["loads", "load", "cfgParser"]

_NUM = re.compile(r"^-?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)$")


class _P:
    def __init__(self, s):
        self.s, self.i, self.n = s, 0, len(s)

    def ws(self):
        s = self.s
        while self.i < self.n:
            c = s[self.i]
            if c in " \t\r\n,":
                self.i += 1
            elif s.startswith("//", self.i) or c == "#":
                while self.i < self.n and s[self.i] != "\n":
                    self.i += 1
            elif s.startswith("/*", self.i):
                j = s.find("*/", self.i + 2)
                self.i = self.n if j < 0 else j + 2
            else:
                break

    def string(self):
        q = self.s[self.i]
        self.i += 1
        out = []
        while self.i < self.n and self.s[self.i] != q:
            c = self.s[self.i]
            if c == "\\" and self.i + 1 < self.n:
                self.i += 1
                c = {"n": "\n", "t": "\t"}.get(self.s[self.i], self.s[self.i])
            out.append(c)
            self.i += 1
        self.i += 1
        return "".join(out)

    def key(self):
        self.ws()
        if self.s[self.i] in "\"'":
            k = self.string()
        else:
            j = self.i
            while self.i < self.n and self.s[self.i] not in ":{}[], \t\r\n":
                self.i += 1
            k = self.s[j:self.i]
        self.ws()
        if self.i >= self.n or self.s[self.i] != ":":
            raise ValueError("expected ':' after key %r at offset %d" % (k, self.i))
        self.i += 1
        return k

    def members(self, closing):
        d = {}
        while True:
            self.ws()
            if self.i >= self.n:
                if closing:
                    raise ValueError("unterminated object")
                return d
            if closing and self.s[self.i] == "}":
                self.i += 1
                return d
            k = self.key()
            d[k] = self.value()

    def value(self):
        self.ws()
        c = self.s[self.i]
        if c == "{":
            self.i += 1
            return self.members(True)
        if c == "[":
            self.i += 1
            out = []
            while True:
                self.ws()
                if self.s[self.i] == "]":
                    self.i += 1
                    return out
                out.append(self.value())
        if c in "\"'":
            return self.string()
        # unquoted scalar: up to end of line / comment / structural character
        j = self.i
        while self.i < self.n and self.s[self.i] not in ",}]\n\r":
            if self.s.startswith("//", self.i) or self.s[self.i] == "#" or self.s.startswith("/*", self.i):
                break
            self.i += 1
        tok = self.s[j:self.i].strip()
        if tok == "true":
            return True
        if tok == "false":
            return False
        if tok == "null":
            return None
        if _NUM.match(tok):
            return float(tok) if any(ch in tok for ch in ".eE") else int(tok)
        return tok


def loads(text):
    p = _P(text)
    p.ws()
    if p.i < p.n and p.s[p.i] == "{":
        p.i += 1
        return p.members(True)
    return p.members(False)


def load(path):
    with open(path, "r") as f:
        return loads(f.read())


class cfgParser:
    """Same accessors as the reference's cfgParser (cfgParser.py:11-74)."""

    def __init__(self, cfg_file=None, contents=None):
        self.contents = load(cfg_file) if contents is None else contents

    def _dataset(self):
        for sec in ("train", "eval"):
            if sec in self.contents and "dataset_name" in self.contents[sec]:
                return self.contents[sec]["dataset_name"]
        return "semantickitti"

    def get_core_vars(self):
        return self.contents["core"]

    def get_train_vars(self):
        return self.contents["train"]

    def get_eval_vars(self):
        return self.contents["eval"]

    def get_model_vars(self):
        return self.contents["model"]

    def get_lattice_gpu_vars(self):
        return self.contents["lattice_gpu"]

    def get_loader_semantic_kitti_vars(self):
        return self.contents["loader_semantic_kitti"]

    def get_loader_paris_lille_vars(self):
        return self.contents["loader_paris_lille"]

    def get_loader_vars(self):
        name = self._dataset()
        if name == "semantickitti":
            return self.get_loader_semantic_kitti_vars()
        if name == "parislille":
            return self.get_loader_paris_lille_vars()
        print("The dataloader you requested is not found: ", name)
        return None

    def get_label_mngr_vars(self):
        return self.get_loader_vars()["label_mngr"]

    def get_transformer_vars(self):
        return self.get_loader_vars()["transformer"]

    def get_visualization_vars(self):
        return self.contents["visualization"]
