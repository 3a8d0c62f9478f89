"""Byte-level BPE detokeniser + output post-strip: restatement of `Qwen3Tokenizer.decode`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows Sources/AudioCommon/Tokenizer.swift:111-181 and Qwen3ASR.swift:283-289.
"""


def byte_to_unicode():
    """Tokenizer.swift:146-172 (GPT-2 table)."""
    keep = list(range(33, 127)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, n = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(0x100 + n)
            n += 1
    return table


_B2U = byte_to_unicode()
_U2B = {c: b for b, c in _B2U.items()}
_WS = " \t\n\r\x0b\x0c               　"
# Foundation's CharacterSet.whitespaces = Unicode General Category Zs + TAB (no newlines).
_WHITESPACES = "\t                　"


def decode(tokens, id_to_token):
    """Tokenizer.swift:111-142."""
    buf = bytearray()
    for tid in tokens:
        tok = id_to_token.get(int(tid))
        if tok is None:
            continue
        if tok.startswith("<|") and tok.endswith("|>"):
            continue
        if tok.startswith("<") and tok.endswith(">") and "|" not in tok:
            buf += tok.encode("utf-8")
            continue
        for ch in tok:
            b = _U2B.get(ch)
            if b is not None:
                buf.append(b)
            else:
                buf += ch.encode("utf-8")
    text = bytes(buf).decode("utf-8", errors="replace")
    return text.strip(_WHITESPACES)


def strip_asr_prefix(raw):
    """Qwen3ASR.swift:285-289."""
    i = raw.find("<asr_text>")
    if i >= 0:
        return raw[i + len("<asr_text>"):].strip(_WHITESPACES)
    return raw


# ---- encode (text -> ids): restatement of Tokenizer.swift:183-289 -------------------------------------------

def _pre_tokenize(text):
    """Tokenizer.swift:220-241: split before every space / newline / tab, the whitespace char starts the next
    word; each word is mapped to its byte-level (GPT-2) representation."""
    words, cur = [], ""
    for ch in text:
        if ch in (" ", "\n", "\t"):
            if cur:
                words.append(cur)
            cur = ch
        else:
            cur += ch
    if cur:
        words.append(cur)
    return ["".join(_B2U[b] for b in w.encode("utf-8")) for w in words]


def _bpe(word, ranks):
    """Tokenizer.swift:244-278: repeatedly merge the lowest-ranked adjacent pair (all its occurrences)."""
    pieces = list(word)
    while len(pieces) > 1:
        best, best_rank = None, None
        for i in range(len(pieces) - 1):
            r = ranks.get(pieces[i] + " " + pieces[i + 1])
            if r is not None and (best_rank is None or r < best_rank):
                best_rank, best = r, (pieces[i], pieces[i + 1])
        if best is None:
            break
        out, i = [], 0
        while i < len(pieces):
            if i < len(pieces) - 1 and pieces[i] == best[0] and pieces[i + 1] == best[1]:
                out.append(best[0] + best[1])
                i += 2
            else:
                out.append(pieces[i])
                i += 1
        pieces = out
    return pieces


def parse_merges(text):
    """Tokenizer.swift:92-106: rank = line index; '#'-prefixed and empty lines are skipped."""
    ranks = {}
    for idx, line in enumerate(text.split("\n")):
        if line.startswith("#") or not line:
            continue
        parts = line.split(" ")
        if len(parts) != 2:
            continue
        ranks[parts[0] + " " + parts[1]] = idx
    return ranks


def encode(text, token_to_id, ranks):
    """Tokenizer.swift:195-217 (+ character fallback :281-289 when no merges are loaded)."""
    if not ranks:
        return [token_to_id[c] for c in text if c in token_to_id]
    ids = []
    for w in _pre_tokenize(text):
        for piece in _bpe(w, ranks):
            if piece in token_to_id:
                ids.append(token_to_id[piece])
    return ids
