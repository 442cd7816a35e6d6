"""A tiny deterministic whitespace/special-token tokenizer used to pin the prompt-building API surface
(build_inputs / completion) against the reference without shipping a real vocabulary."""
import re

SPECIALS = ["<im_patch>", "<vi_frame>", "<im_start>", "<im_end>", "<vi_start>", "<vi_end>"]


class FakeTokenizer:
    def __init__(self, vocab_text=300):
        self.vocab_text = vocab_text
        self.special = {}
        self.padding_side = "right"
        self.pad_id = 0

    def __len__(self):
        return self.vocab_text + len(self.special)

    def add_tokens(self, toks, special_tokens=False):
        n = 0
        for t in toks:
            if t not in self.special:
                self.special[t] = self.vocab_text + len(self.special)
                n += 1
        return n

    def convert_tokens_to_ids(self, toks):
        if isinstance(toks, str):
            return self.special[toks]
        return [self.special[t] for t in toks]

    def _encode(self, text):
        pat = "(" + "|".join(re.escape(s) for s in self.special) + ")" if self.special else None
        parts = re.split(pat, text) if pat else [text]
        ids = [1]
        for p in parts:
            if not p:
                continue
            if p in self.special:
                ids.append(self.special[p])
            else:
                for w in p.split():
                    ids.append(3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (self.vocab_text - 3)))
        return ids

    def __call__(self, texts, padding=True):
        if isinstance(texts, str):                         # tokenizer(prompt).input_ids -> flat list (HF behaviour)
            from types import SimpleNamespace
            return SimpleNamespace(input_ids=self._encode(texts))
        seqs = [self._encode(t) for t in texts]
        L = max(len(s) for s in seqs)
        if self.padding_side == "left":
            seqs = [[self.pad_id] * (L - len(s)) + s for s in seqs]
        else:
            seqs = [s + [self.pad_id] * (L - len(s)) for s in seqs]
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=seqs)

    eos_token_id = 2

    def decode(self, ids, skip_special_tokens=True):
        return self.batch_decode([ids], skip_special_tokens)[0]

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(f"w{int(i)}" for i in row if not (skip_special_tokens and int(i) >= self.vocab_text)) for row in ids]
