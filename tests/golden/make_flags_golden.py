"""Extracts the reference's flag surface (train.py:45-120) with ast -- train.py itself cannot be
imported here (dgl, tensorboard) -- into tests/golden/train_flags.json."""
import ast
import json
import os

src = open("/root/reference/train.py").read()
flags = {}
for node in ast.walk(ast.parse(src)):
    if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
        name = node.args[0].value
        kw = {}
        for k in node.keywords:
            if k.arg == "type":
                kw["type"] = k.value.id
            elif k.arg in ("default", "action", "nargs"):
                try:
                    kw[k.arg] = ast.literal_eval(k.value)
                except Exception:
                    kw[k.arg] = ast.unparse(k.value)
            elif k.arg == "choices":
                try:
                    kw["choices"] = ast.literal_eval(k.value)
                except Exception:
                    kw["choices"] = "expr"
        flags[name] = kw
json.dump(flags, open(os.path.join(os.path.dirname(__file__), "train_flags.json"), "w"), indent=1, sort_keys=True)
print(len(flags), "flags")
