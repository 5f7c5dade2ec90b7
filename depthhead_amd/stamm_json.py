"""Importer for the reference's serialised model.

`serde_json::to_string(&HoughPrediction)` (examples/hough_tree_trainer.rs:60-67, 183) produces

    {"stepwidth": u32, "subimage_width": u32, "subimage_height": u32, "gaussian_sigma": f32,
     "forest": <stamm RandomForest>, "meanshift_iterations": u32}           (prediction.rs:239-256)

with, inside the forest, payloads whose shapes ARE known from in-tree derives:

    NodeParam  {"r1": Rect, "r2": Rect, "threshold": f64}                    (houghforest.rs:63-68)
    Rect       {"topleft": [x, y], "bottomright": [x2, y2]}                  (types.rs:33-37)
    LeafParam  {"prob": f64, "offsets": [[f32;3]...], "rotations": [[f64;3]...]}   (houghforest.rs:73-78)

UNVERIFIED (parity unpinned): how stamm 0.2.0 (not vendored, Cargo.lock:1154-1162) nests those
payloads -- the key of the tree list, the node container's field names, and which child
`Binar::One` selects.  The importer therefore does not hard-code a schema: it finds the payloads by
their known keys and recovers the tree shape from the JSON nesting (any chain of single-key wrapper
objects, a node object = one NodeParam payload + exactly two child subtrees).  The two children are
taken in document order unless their keys say otherwise (left/right, zero/one, false/true, 0/1);
`one_child` states which of them `Binar::One` follows; it is a REQUIRED argument of `import_json`
(SURVEY.md Appendix B believes "Binar::Zero -> left, Binar::One -> right", unverified -- so no default
is offered, and `flip_suspect` gives a sanity check on a labelled frame).  A flat layout (nodes in an array
referring to each other by index) is accepted too.  `export_json` writes the nested layout this
importer assumes, so a converted model can be inspected and round-tripped.
"""
from __future__ import annotations

import json

import numpy as np

from .forest import NODE_DTYPE, Forest
from .synth import ModelParams

_NODE_KEYS = {"r1", "r2", "threshold"}
_LEAF_KEYS = {"prob", "offsets", "rotations"}
_FIRST_HINTS = ("left", "zero", "false", "lo", "no", "0", "first", "a")
_SECOND_HINTS = ("right", "one", "true", "hi", "yes", "1", "second", "b")


def _is_node_payload(v) -> bool:
    return isinstance(v, dict) and _NODE_KEYS <= set(v)


def _is_leaf_payload(v) -> bool:
    return isinstance(v, dict) and _LEAF_KEYS <= set(v)


def _contains_payload(v) -> bool:
    if _is_node_payload(v) or _is_leaf_payload(v):
        return True
    if isinstance(v, dict):
        return any(_contains_payload(x) for x in v.values())
    if isinstance(v, list):
        return any(_contains_payload(x) for x in v)
    return False


def _unwrap_node_payload(v):
    """Follow single-entry wrappers (enum tags, newtypes) down to a NodeParam payload that holds no
    further payloads; None if `v` is something else (e.g. a child subtree)."""
    while True:
        if _is_node_payload(v):
            return v if not any(_contains_payload(y) for k, y in v.items() if k not in _NODE_KEYS) else None
        if isinstance(v, dict) and len(v) == 1:
            v = next(iter(v.values()))
        elif isinstance(v, list) and len(v) == 1:
            v = v[0]
        else:
            return None


def _rect(r) -> tuple:
    (x0, y0), (x1, y1) = r["topleft"], r["bottomright"]
    return int(x0), int(y0), int(x1), int(y1)


class _Builder:
    def __init__(self, one_child: str):
        if one_child not in ("left", "right"):
            raise ValueError("one_child must be 'left' or 'right'")
        self.one_is_second = one_child == "right"
        self.nodes, self.leaf_prob, self.offs, self.rots = [], [], [], []

    def leaf(self, p) -> int:
        self.leaf_prob.append(float(p["prob"]))
        self.offs.append(np.asarray(p["offsets"], dtype=np.float32).reshape(-1, 3))
        self.rots.append(np.asarray(p["rotations"], dtype=np.float64).reshape(-1, 3))
        return ~(len(self.leaf_prob) - 1)

    def subtree(self, v) -> int:
        """Returns a child reference (node index >= 0, or ~leaf)."""
        while True:   # unwrap tags / single-field wrappers / Option-like one-element lists
            if _is_leaf_payload(v):
                return self.leaf(v)
            if isinstance(v, dict) and not _is_node_payload(v):
                inner = [(k, x) for k, x in v.items() if _contains_payload(x)]
                if len(inner) == 1:
                    v = inner[0][1]
                    continue
            if isinstance(v, list):
                inner = [x for x in v if _contains_payload(x)]
                if len(inner) == 1:
                    v = inner[0]
                    continue
            break
        payload, kids = None, []
        items = list(v.items()) if isinstance(v, dict) else [(str(i), x) for i, x in enumerate(v)] if isinstance(v, list) else []
        if _is_node_payload(v):
            payload = v
            items = [(k, x) for k, x in items if k not in _NODE_KEYS]
        for k, x in items:
            if payload is None and _unwrap_node_payload(x) is not None:
                payload = _unwrap_node_payload(x)
            elif _contains_payload(x):
                kids.append((k, x))
        if payload is None or len(kids) != 2:
            raise ValueError(f"cannot recover a split node: {len(kids)} child subtrees, payload={'yes' if payload else 'no'}")
        (k0, c0), (k1, c1) = kids
        l0, l1 = k0.lower(), k1.lower()
        if any(h == l0 or l0.startswith(h) for h in _SECOND_HINTS) and any(h == l1 or l1.startswith(h) for h in _FIRST_HINTS):
            c0, c1 = c1, c0
        idx = len(self.nodes)
        self.nodes.append(None)
        first, second = self.subtree(c0), self.subtree(c1)
        zero, one = (first, second) if self.one_is_second else (second, first)
        self.nodes[idx] = (_rect(payload["r1"]), _rect(payload["r2"]), float(payload["threshold"]), zero, one)
        return idx

    def flat_tree(self, arr, root) -> int:
        """Nodes in an array that refer to each other by index: an entry is a leaf payload, or a
        NodeParam payload (possibly nested one level) plus two integer fields."""
        memo = {}

        def rec(i):
            if i in memo:
                raise ValueError("node referenced twice")
            e = arr[i]
            if _is_leaf_payload(e):
                memo[i] = self.leaf(e)
                return memo[i]
            inner = [x for x in e.values() if _is_leaf_payload(x)] if isinstance(e, dict) else []
            ints = [(k, x) for k, x in e.items() if isinstance(x, int) and not isinstance(x, bool)] if isinstance(e, dict) else []
            payload = e if _is_node_payload(e) else next((x for x in e.values() if _is_node_payload(x)), None)
            if payload is None and len(inner) == 1:
                memo[i] = self.leaf(inner[0])
                return memo[i]
            if payload is None or len(ints) != 2:
                raise ValueError("unrecognised flat node entry")
            (k0, a), (k1, b) = ints
            if any(k0.lower().startswith(h) for h in _SECOND_HINTS) and any(k1.lower().startswith(h) for h in _FIRST_HINTS):
                a, b = b, a
            idx = len(self.nodes)
            self.nodes.append(None)
            memo[i] = idx
            first, second = rec(a), rec(b)
            zero, one = (first, second) if self.one_is_second else (second, first)
            self.nodes[idx] = (_rect(payload["r1"]), _rect(payload["r2"]), float(payload["threshold"]), zero, one)
            return idx

        return rec(root)

    def forest(self, roots) -> Forest:
        nodes = np.zeros(len(self.nodes), dtype=NODE_DTYPE)
        for i, n in enumerate(self.nodes):
            nodes[i] = n
        begin_o = np.zeros(len(self.leaf_prob) + 1, dtype=np.uint32)
        begin_r = np.zeros(len(self.leaf_prob) + 1, dtype=np.uint32)
        np.cumsum([len(o) for o in self.offs], out=begin_o[1:])
        np.cumsum([len(r) for r in self.rots], out=begin_r[1:])
        offs = np.concatenate(self.offs) if self.offs and begin_o[-1] else np.zeros((0, 3), np.float32)
        rots = np.concatenate(self.rots) if self.rots and begin_r[-1] else np.zeros((0, 3), np.float64)
        return Forest(np.asarray(roots, dtype=np.int32), nodes, np.asarray(self.leaf_prob), begin_o, begin_r, offs, rots)


def _find_tree_list(v):
    """The first list (depth-first, document order) whose items each contain a payload."""
    if isinstance(v, list) and v and all(_contains_payload(x) for x in v) and not all(_is_leaf_payload(x) or _is_node_payload(x) for x in v):
        return v
    if isinstance(v, dict):
        for x in v.values():
            r = _find_tree_list(x)
            if r is not None:
                return r
    if isinstance(v, list):
        for x in v:
            r = _find_tree_list(x)
            if r is not None:
                return r
    return None


def _flat_nodes(tree):
    """(array, root index) if `tree` looks like the flat layout, else None."""
    if not isinstance(tree, dict):
        return None
    for k, v in tree.items():
        if isinstance(v, list) and v and all(isinstance(e, dict) for e in v) and \
                any(isinstance(x, int) and not isinstance(x, bool) for e in v for x in e.values()) and \
                all(_contains_payload(e) for e in v):
            root = next((x for kk, x in tree.items() if isinstance(x, int) and not isinstance(x, bool) and "root" in kk.lower()), 0)
            return v, root
    return None


def import_json(text: str, *, one_child: str) -> tuple[Forest, ModelParams]:
    """JSON of a serialised `HoughPrediction` -> (Forest, ModelParams).

    `one_child` ("left" / "right": the child `Binar::One` selects) has NO default: stamm 0.2.0 is not
    vendored, so the convention cannot be verified here, and a silently flipped import yields
    plausible but wrong poses.  The caller states it; `check_child_convention` below helps to tell
    the two apart on a labelled frame.  PARITY UNPINNED (stamm layout / child order)."""
    if one_child not in ("left", "right"):
        raise ValueError("one_child must be 'left' or 'right' (which child Binar::One selects; unverifiable here -- see INTEGRATION.md)")
    doc = json.loads(text)
    try:
        params = ModelParams(doc["stepwidth"], doc["subimage_width"], doc["subimage_height"], doc["gaussian_sigma"],
                             doc["meanshift_iterations"])
        forest_doc = doc["forest"]
    except (KeyError, TypeError) as e:
        raise ValueError(f"not a serialised HoughPrediction (prediction.rs:239-256): missing {e}") from None
    trees = _find_tree_list(forest_doc)
    if trees is None:
        raise ValueError("no list of trees found inside 'forest'")
    b = _Builder(one_child)
    roots = []
    for tree in trees:
        flat = _flat_nodes(tree)
        roots.append(b.flat_tree(*flat) if flat else b.subtree(tree))
    return b.forest(roots), params


def export_json(forest: Forest, params: ModelParams, one_child: str = "right") -> str:
    """Writes the nested layout `import_json` assumes (see the module docstring; NOT verified to be
    byte-compatible with stamm's own serialisation)."""
    cz, co = forest.nodes["child_zero"], forest.nodes["child_one"]

    def rect(r):
        return {"topleft": [int(r[0]), int(r[1])], "bottomright": [int(r[2]), int(r[3])]}

    def leaf(i):
        o = forest.offsets[forest.off_begin[i]:forest.off_begin[i + 1]]
        r = forest.rotations[forest.rot_begin[i]:forest.rot_begin[i + 1]]
        return {"Leaf": {"prob": float(forest.leaf_prob[i]), "offsets": [[float(x) for x in v] for v in o],
                         "rotations": [[float(x) for x in v] for v in r]}}

    def sub(ref):
        stack_out = {}
        # iterative post-order to survive deep trees
        order, stack = [], [ref]
        while stack:
            n = stack.pop()
            order.append(n)
            if n >= 0:
                stack.extend([int(cz[n]), int(co[n])])
        for n in reversed(order):
            if n < 0:
                stack_out[n] = leaf(~n)
            else:
                nd = forest.nodes[n]
                zero, one = stack_out[int(cz[n])], stack_out[int(co[n])]
                left, right = (zero, one) if one_child == "right" else (one, zero)
                stack_out[n] = {"Inner": {"param": {"r1": rect(nd["r1"]), "r2": rect(nd["r2"]), "threshold": float(nd["threshold"])},
                                          "left": left, "right": right}}
        return stack_out[ref]

    doc = {"stepwidth": params.stepwidth, "subimage_width": params.subimage_width, "subimage_height": params.subimage_height,
           "gaussian_sigma": params.gaussian_sigma,
           "forest": {"subtrees": [{"root": sub(int(r))} for r in forest.roots]},
           "meanshift_iterations": params.meanshift_iterations}
    return json.dumps(doc)


def flip_suspect(poses_as_imported, poses_flipped, truth_mid, tol_mm: float = 150.0) -> bool:
    """Sanity check against a flipped `Binar::One` convention on labelled frames (e.g. BIWI ground
    truth, `depthhead_amd.biwi`): given the head positions predicted with the model as imported and
    with `one_child` flipped, returns True when the FLIPPED import lands within `tol_mm` of the
    ground truth on more frames than the import under test -- i.e. the caller's `one_child` is
    probably wrong.  A heuristic for humans, not a parity statement."""
    a = np.linalg.norm(np.asarray(poses_as_imported, dtype=np.float64) - np.asarray(truth_mid, dtype=np.float64), axis=-1)
    b = np.linalg.norm(np.asarray(poses_flipped, dtype=np.float64) - np.asarray(truth_mid, dtype=np.float64), axis=-1)
    return int((b <= tol_mm).sum()) > int((a <= tol_mm).sum())
