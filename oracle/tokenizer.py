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
