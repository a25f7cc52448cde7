"""Text normalisation in front of ``transcript2phonemids``.

The reference runs NeMo text normalisation + uroman romanisation (normalize.py:28-47); both are optional
third-party host dependencies and out of scope for the GPU path.  When they are
importable they are used exactly like the reference; otherwise only the reference's final regex steps are
applied to the lower-cased text (numbers etc. are then NOT expanded -- a documented limitation).
"""
import re


def _third_party(lang):
    try:
        import uroman                                                    # noqa: F401
        from nemo_text_processing.text_normalization.normalize import Normalizer
    except Exception:
        return None
    return uroman.Uroman(), Normalizer(input_case="cased", lang=lang)


class ZeroVoxNormalizer:
    def __init__(self, lang):
        self._lang = lang
        self._tp = None
        self._probed = False

    @property
    def language(self):
        return self._lang

    def normalize(self, transcript):
        if not self._probed:
            self._tp, self._probed = _third_party(self._lang), True
        if self._tp is not None:
            uromanizer, nemo = self._tp
            transcript = str(uromanizer.romanize_string(nemo.normalize(transcript)))
        uro = transcript.lower().strip()
        norm = re.sub("([^a-z' ])", " ", uro)                            # normalize.py:38-40
        norm = re.sub(" +", " ", norm).strip()
        return uro, norm
