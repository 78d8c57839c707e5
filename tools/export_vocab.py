"""Export the reference's kana inventory (data, not code) to chainer-speech-recognition_amd/asr/vocab_tables.json.
usage: python tools/export_vocab.py /path/to/reference   (needs only asr/vocab.py of that checkout)"""
import importlib.util, json, os, sys
ref = sys.argv[1]
spec = importlib.util.spec_from_file_location("ref_vocab", os.path.join(ref, "asr/vocab.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chainer-speech-recognition_amd", "asr", "vocab_tables.json")
json.dump(dict(unigram_tokens=list(m.UNIGRAM_TOKENS), sutegana=list(m.SUTEGANA), collapse=dict(m.UNIGRAM_COLLAPSE), blank=m.ID_BLANK),
          open(out, "w", encoding="utf-8"), ensure_ascii=False, indent=0)
print("wrote", out)
