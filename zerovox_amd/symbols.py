"""Phone / punctuation id maps -- same contract as the reference ``Symbols`` (symbols.py:2-48)."""


class Symbols:
    NO_PUNCT = "_NP_"

    def __init__(self, phones, puncts):
        # phones are numbered from 0 in order; punctuation from 1 (0 = NO_PUNCT)   symbols.py:6-22
        self._phone2id = {p: i for i, p in enumerate(phones)}
        self._id2phone = {i: p for p, i in self._phone2id.items()}
        self._punct2id = {Symbols.NO_PUNCT: 0}
        for i, p in enumerate(puncts):
            self._punct2id[p] = i + 1
        self._id2punct = {i: p for p, i in self._punct2id.items()}

    def is_phone(self, p):
        return p in self._phone2id

    def encode_phone(self, phone):
        return self._phone2id[phone]

    def decode_phone(self, phone):
        return self._id2phone[phone]

    @property
    def num_phones(self):
        return len(self._phone2id)

    def is_punct(self, p):
        return p in self._punct2id

    def encode_punct(self, punct):
        return self._punct2id[punct]

    def decode_punct(self, punct):
        return self._id2punct[punct]

    @property
    def num_puncts(self):
        return len(self._punct2id)
