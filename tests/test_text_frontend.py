"""§8(f) item 1: vocabulary + text_to_sequence + collate against outputs captured from the reference."""
import json
import os

import numpy as np

import cmtts_amd
from cmtts_amd import text
from conftest import GOLDEN_DIR


def test_text_to_sequence_matches_reference():
    g = json.load(open(os.path.join(GOLDEN_DIR, "text.json")))
    assert len(text.symbols) == 360 and text.symbols[0] == "_"
    for line, ids in zip(g["lines"], g["ids"]):
        assert text.text_to_sequence(line.split("|")[2], []) == ids
    assert max(max(i) for i in g["ids"]) < cmtts_amd.get_config("LJSpeech").n_symbols


def test_collate_shapes_and_padding(tmp_path):
    g = json.load(open(os.path.join(GOLDEN_DIR, "text.json")))
    p = tmp_path / "val.txt"
    p.write_text("\n".join(g["lines"]) + "\n")
    names, speakers, texts, raws = text.read_meta(str(p))
    items = [(n, 0, text.text_to_sequence(t), r, np.ones((1, 512), np.float32) * i)
             for i, (n, t, r) in enumerate(zip(names, texts, raws))]
    ids, raw, spk, padded, lens, L, emb = text.collate(items, load_spker_embed=True)
    assert ids == names and L == max(len(i) for i in g["ids"]) and padded.shape == (4, L)
    for row, ref in zip(padded, g["ids"]):
        assert row[: len(ref)].tolist() == ref and (row[len(ref):] == 0).all()
    assert emb.shape == (4, 512) and emb[2, 0] == 2.0 and lens.tolist() == [len(i) for i in g["ids"]]
