"""Text -> phoneme-ID frontend and batch collation: the part of the reference's
`text/` package and `TextDataset` the inference path needs to consume `val.txt`-format lines
(`name|speaker|{ARPAbet}|raw`, dataset.py:271-283).  G2P / English normalisation stay out of scope
(they need g2p_en, inflect, unidecode and the missing lexicon blob); only the no-op and whitespace/
lower-case cleaners exist here.

`symbols.json` is the reference's 360-entry vocabulary (text/symbols.py:21-29) exported as data by
tests/golden/make_golden.py; embedding row = index, row 0 = padding.
"""
import json
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "symbols.json")) as _f:
    symbols = json.load(_f)
_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_curly_re = re.compile(r"(.*?)\{(.+?)\}(.*)")
_whitespace_re = re.compile(r"\s+")


def basic_cleaners(text):
    return _whitespace_re.sub(" ", text.lower())


_CLEANERS = {"basic_cleaners": basic_cleaners}


def _clean(text, cleaner_names):
    for name in cleaner_names or []:
        if name not in _CLEANERS:
            raise NotImplementedError(f"cleaner {name!r} needs unidecode/inflect (absent): pass phonemes in {{braces}}")
        text = _CLEANERS[name](text)
    return text


def _keep(s):
    return s in _symbol_to_id and s != "_" and s != "~"


def text_to_sequence(text, cleaner_names=None):
    """text/__init__.py:15-41: characters map to their own symbols, `{...}` spans are ARPAbet
    (looked up with the `@` prefix); unknown symbols are silently dropped."""
    seq = []
    while len(text):
        m = _curly_re.match(text)
        if not m:
            seq += [_symbol_to_id[s] for s in _clean(text, cleaner_names) if _keep(s)]
            break
        seq += [_symbol_to_id[s] for s in _clean(m.group(1), cleaner_names) if _keep(s)]
        seq += [_symbol_to_id["@" + s] for s in m.group(2).split() if _keep("@" + s)]
        text = m.group(3)
    return seq


def read_meta(path):
    """TextDataset.process_meta dataset.py:271-283."""
    names, speakers, texts, raws = [], [], [], []
    with open(path, encoding="utf-8") as f:
        for line in f:
            n, s, t, r = line.strip("\n").split("|")
            names.append(n); speakers.append(s); texts.append(t); raws.append(r)
    return names, speakers, texts, raws


def collate(items, load_spker_embed=False):
    """TextDataset.collate_fn dataset.py:285-296.  items: (basename, speaker_id, phone ids, raw_text,
    spker_embed[1,512] | None) -> the 7-tuple CMTotalTTSSynthesize.synthesize takes."""
    ids = [d[0] for d in items]
    speakers = np.array([d[1] for d in items])
    texts = [np.asarray(d[2], np.int64) for d in items]
    raw_texts = [d[3] for d in items]
    text_lens = np.array([t.shape[0] for t in texts])
    spk = np.concatenate([np.asarray(d[4], np.float32).reshape(1, -1) for d in items], 0) if load_spker_embed else None
    L = int(text_lens.max())
    padded = np.zeros((len(texts), L), np.int64)
    for i, t in enumerate(texts):
        padded[i, : t.shape[0]] = t
    return ids, raw_texts, speakers, padded, text_lens, L, spk
