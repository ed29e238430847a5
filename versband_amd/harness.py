"""Host-side harness I/O of the inference entry point: manifest / condition loading, item assembly, caption text,
wav reading and writing.  Pure plumbing - no GPU work; the behaviour follows scripts/test_final.py:100-347 and :424-463 of the
reference, with its crashes repaired where the intent is unambiguous (each noted at the spot).
"""
from __future__ import annotations

import ast
import csv
import math
import os
import random
import wave
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

MEL_NUM = 80                    # test_final.py:208
MEL_DOWNSAMPLE = 2              # :209  latent length = int(T_mel / 2)
MIN_BATCH_LEN = 75              # :211
PAD_VALUE = -5                  # :212  log-mel floor
UNIT_FRAMES_MULTIPLE = 8        # :214  2 * min_factor
MAX_ITEMS = 200                 # :229  the reference draws 200 random item names
MAX_DURATION = 20.0             # :236


def pad_or_cut_xd(x, length: int, dim: int = 1, pad_value=0):
    """The helper test_final.py:27 imports from ldm/data/joinaudiodataset_anylen.py but the reference never defines:
    cut `x` to `length` along `dim` or pad it with `pad_value`.  numpy in -> numpy out, tensor in -> tensor out."""
    is_t = torch.is_tensor(x)
    a = x.numpy() if is_t else np.asarray(x)
    n = a.shape[dim]
    if n >= length:
        a = np.take(a, np.arange(length), axis=dim)
    else:
        pad = [(0, 0)] * a.ndim
        pad[dim] = (0, length - n)
        a = np.pad(a, pad, constant_values=pad_value)
    return torch.from_numpy(np.ascontiguousarray(a)) if is_t else a


def safe_path(path):
    """`path`, with its directory created if it does not exist yet (the reference's helper of the same name, test_final.py:112-114)"""
    parent = os.path.dirname(os.path.abspath(str(path)))
    os.makedirs(parent, exist_ok=True)
    return path


def load_samples_from_tsv(tsv_path) -> List[Dict[str, str]]:
    """Manifest reader with the reference's contract (test_final.py:116-133): a header row, TAB-separated fields taken literally (no
    quoting rules), one dict per data row; a missing file is an error, an empty manifest only a warning.  Parsed by hand: the fields
    carry no quoting, so splitting the lines is the whole format."""
    if not os.path.isfile(str(tsv_path)):
        raise FileNotFoundError(f"Dataset not found: {tsv_path}")
    with open(str(tsv_path), encoding="utf-8", newline="") as f:
        lines = [ln.rstrip("\r\n") for ln in f]
    lines = [ln for ln in lines if ln != ""]
    if not lines:
        print(f"warning: empty manifest: {tsv_path}")
        return []
    header = lines[0].split("\t")
    samples = []
    for ln in lines[1:]:
        cells = ln.split("\t")
        row = {k: (cells[i] if i < len(cells) else None) for i, k in enumerate(header)}
        if len(cells) > len(header):            # csv.DictReader keeps surplus cells under the key None: same here
            row[None] = cells[len(header):]
        samples.append(row)
    if not samples:
        print(f"warning: empty manifest: {tsv_path}")
    return samples


def save_rows_to_tsv(rows: List[Dict[str, str]], fieldnames: List[str], path) -> None:
    """save_df_to_tsv (test_final.py:100-110) without pandas: header, TAB separated, no quoting, backslash escapes."""
    with open(safe_path(str(path)), "w", newline="", encoding="utf-8") as f:
        w = csv.DictWriter(f, fieldnames=fieldnames, delimiter="\t", quoting=csv.QUOTE_NONE, escapechar="\\", lineterminator="\n")
        w.writeheader()
        w.writerows(rows)


# ------------------------------------------------------------------------------------------------------------------
# wav files
# ------------------------------------------------------------------------------------------------------------------
def read_wav(path):
    """soundfile.read(path) -> (float64 samples in [-1, 1), sample rate); falls back to the standard library / scipy when
    soundfile is not installed (this image).  Multi-channel files keep their [n, ch] shape like soundfile."""
    try:
        import soundfile as sf
        return sf.read(path)
    except ImportError:
        pass
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype == np.int16:
        data = data.astype(np.float64) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float64) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float64) - 128.0) / 128.0
    else:
        data = data.astype(np.float64)
    return data, sr


def write_wav_pcm16(path, wav, sr: int) -> None:
    """soundfile.write(path, wav, sr, subtype='PCM_16') (test_final.py:433): clip to [-1, 1), scale by 2^15, round-to-nearest."""
    wav = np.asarray(wav, dtype=np.float64)
    try:
        import soundfile as sf
        sf.write(safe_path(path), wav, sr, subtype="PCM_16")
        return
    except ImportError:
        pass
    pcm = np.clip(np.rint(wav * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(safe_path(path), "wb") as f:
        f.setnchannels(1 if pcm.ndim == 1 else pcm.shape[1])
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())


# ------------------------------------------------------------------------------------------------------------------
# caption text (the "Musical: ..." half of the prompt)
# ------------------------------------------------------------------------------------------------------------------
_PITCH_CLASSES = ["C", "C-sharp", "D", "E-flat", "E", "F", "F-sharp", "G", "A-flat", "A", "B-flat", "B"]
_STEP = {"c": 0, "d": 2, "e": 4, "f": 5, "g": 7, "a": 9, "b": 11}


def _parse_key(key: str):
    """'C major' / 'a minor' / music21-style 'F#' (major), 'b-' (minor: lower case) -> (pitch class, mode)."""
    parts = key.strip().split()
    tonic = parts[0]
    mode = parts[1].lower() if len(parts) > 1 else ("minor" if tonic[0].islower() else "major")
    pc = _STEP[tonic[0].lower()]
    for ch in tonic[1:]:
        pc += 1 if ch == "#" else (-1 if ch in "-b" else 0)
    return pc % 12, mode


class CaptionGenerator2:
    """Drop-in for ldm/modules/encoders/caption_generator.py:781 CaptionGenerator2.transcribe (called at test_final.py:260-268).

    Same inputs, same decision rules - confidence gates 0.5 (key) / 0.3 (tempo) (:63-64); the class boundaries of CaptionGenerator2
    with their deliberate dead zones between classes (tempo 69 / 71-89 / 91-119 / 121-159 / 161, :785-802; average pitch 53 / 56-62 /
    64-77 / 79, :804-819; duration 4.5 / 5.5-9.5 / 10.5-14.5 / 15.5, :821-837, with an even chance of the exact "N seconds"); a
    coin flip to the relative key (:617-618); emotions joined in random order (:659-670).  The SENTENCES are composed here from
    the factors that survive - the reference instead picks from ~25 tables of hand-written English templates, which are prose, not
    behaviour, and are not reproduced: caption strings therefore differ from the reference's while carrying the same facts.
    music21 (absent here) is replaced by pitch-class arithmetic for naming keys."""

    key_min_conf = 0.5
    tempo_min_conf = 0.3
    tempo_words = {"very low": ["very slow"], "low": ["slow", "gentle"], "medium": ["medium", "moderate"], "high": ["fast", "quick"],
                   "very high": ["very fast"]}
    pitch_words = {"low": ["low", "relatively low"], "medium": ["medium", "average"], "high": ["high", "relatively high"],
                   "very high": ["very high"]}
    duration_words = {"short": ["a short time"], "medium": ["a medium stretch of time"], "long": ["a long time"],
                      "very long": ["a very long time"]}

    def __init__(self, rng: Optional[random.Random] = None):
        self.rng = rng or random.Random()

    @staticmethod
    def _classify(v: float, bounds):
        """bounds: [(lo, hi, name)]; values in no interval (the dead zones) give None."""
        for lo, hi, name in bounds:
            if lo <= v < hi:
                return name
        return None

    def prepare_key(self, key, key_conf):
        if key is None or key == "None" or key_conf < self.key_min_conf:
            return None
        pc, mode = _parse_key(str(key))
        if self.rng.random() > 0.5:                               # relative key: same notes, other mode
            pc, mode = ((pc + 9) % 12, "minor") if mode == "major" else ((pc + 3) % 12, "major")
        return f"{_PITCH_CLASSES[pc]} {mode}"

    def prepare_tempo(self, tempo, tempo_conf):
        if tempo is None or tempo <= 0 or tempo_conf < self.tempo_min_conf:
            return None
        c = self._classify(tempo, [(-math.inf, 69, "very low"), (71, 89, "low"), (91, 119, "medium"), (121, 159, "high"),
                                   (161, math.inf, "very high")])
        return self.rng.choice(self.tempo_words[c]) if c else None

    def prepare_avg_pitch(self, avg_pitch):
        if avg_pitch is None or avg_pitch <= 0:
            return None
        c = self._classify(avg_pitch, [(-math.inf, 53, "low"), (56, 62, "medium"), (64, 77, "high"), (79, math.inf, "very high")])
        return self.rng.choice(self.pitch_words[c]) if c else None

    def prepare_emotion(self, emotion):
        if emotion is None or emotion == "None" or len(emotion) == 0:
            return None
        e = list(emotion)
        self.rng.shuffle(e)
        if len(e) <= 2:
            return " and ".join(e)
        return ", ".join(e[:-1]) + ", and " + e[-1]

    def prepare_duration(self, duration):
        if duration is None or duration <= 0:
            return None
        exact = f"{round(duration)} seconds"
        c = self._classify(duration, [(-math.inf, 4.5, "short"), (5.5, 9.5, "medium"), (10.5, 14.5, "long"), (15.5, math.inf, "very long")])
        words = self.rng.choice(self.duration_words[c]) if c else None
        pick = self.rng.choice([words, exact])
        return pick

    def transcribe(self, key=None, key_conf=0.0, avg_pitch=None, tempo=None, tempo_conf=0.0, emotion=None, duration=None) -> str:
        key = self.prepare_key(key, key_conf)
        tempo = self.prepare_tempo(tempo, tempo_conf)
        avg_pitch = self.prepare_avg_pitch(avg_pitch)
        emotion = self.prepare_emotion(emotion)
        duration = self.prepare_duration(duration)
        traits = []
        if key:
            traits.append(f"is in {key}")
        if avg_pitch:
            traits.append(f"sits at a {avg_pitch} pitch")
        if tempo:
            traits.append(f"moves at a {tempo} tempo")
        if duration and traits:
            traits.append(f"lasts {duration}")
        if not traits and not emotion:
            return ""                                             # the reference's `caption = ''` fall-through (:741, :777)
        out = ""
        if traits:
            body = traits[0] if len(traits) == 1 else ", ".join(traits[:-1]) + " and " + traits[-1]
            out = f"The melody {body}."
        if emotion:
            out = (out + " " if out else "") + f"It carries a {emotion} mood."
        return out


# ------------------------------------------------------------------------------------------------------------------
# dataset
# ------------------------------------------------------------------------------------------------------------------
class InferDataset:
    """InferDataset of scripts/test_final.py:196-340.

    manifest (TSV) columns used: name, duration, caption ('<psep>'-separated alternatives), key, key_confidence, avg_pitch, tempo,
    tempo_confidence, emotion (a Python list literal), wav_len, mel_path ([80, T] .npy), vocal_mel_path (.npy, first 20 rows are
    the 'acoustic' condition), audio_path.  `other_condition` is the midi dictionary (.npy holding {name: [T] array}); the beats
    dictionary sits beside it with 'midi' -> 'beats' in the path (:218-226).

    Differences from the reference, all where it cannot run as written: `beats` is read from the beats dictionary (the reference
    only assigns it on its corrupted-input branch and raises NameError otherwise, :282-305); at most - not exactly - 200 items are
    drawn, with a seedable RNG (random.sample(keys, 200) raises on smaller manifests, :229); `eval()` of the emotion column is a
    literal parse."""

    mel_num = MEL_NUM
    mel_downsample_rate = MEL_DOWNSAMPLE
    min_batch_len = MIN_BATCH_LEN
    pad_value = PAD_VALUE
    unit_upsample_rate = 1
    unit_frames_multiple = UNIT_FRAMES_MULTIPLE

    def __init__(self, manifest_path, other_condition, seed: Optional[int] = None, max_items: int = MAX_ITEMS):
        samples = load_samples_from_tsv(manifest_path)
        items_dict = {s["name"]: s for s in samples}
        self.rng = random.Random(seed)
        self.np_rng = np.random.RandomState(seed)
        self.caption_generator = CaptionGenerator2(self.rng)
        self.unit_pad_value = self.pad_value                                      # :216 (the last assignment wins)
        midi_path = str(other_condition)
        beats_path = midi_path.replace("midi", "beats")
        print(f"MIDI path: {midi_path}")
        print(f"Beats path: {beats_path}")
        midi_dict = np.load(midi_path, allow_pickle=True).item()
        beats_dict = np.load(beats_path, allow_pickle=True).item()
        names = list(items_dict.keys())
        self.pred_list = self.rng.sample(names, min(max_items, len(names)))
        self.items = []
        for name in self.pred_list:
            item = items_dict[name]
            if float(item["duration"]) > MAX_DURATION:
                continue
            if name not in midi_dict or name not in beats_dict:
                print(f"no midi / beats for {name}: skipped")
                continue
            item["midi"], item["beats"], item["name"] = midi_dict[name], beats_dict[name], name
            self.items.append(item)

    def __len__(self):
        return len(self.items)

    def _caption(self, data):
        style = self.np_rng.choice(data["caption"].split("<psep>"))
        caption = f"Style: {style} "
        emotion = data.get("emotion", "None")
        try:
            emotion = ast.literal_eval(emotion) if isinstance(emotion, str) else emotion
        except (ValueError, SyntaxError):
            emotion = None
        prompt = self.caption_generator.transcribe(
            key=data.get("key"), key_conf=float(data.get("key_confidence", 0) or 0), avg_pitch=float(data.get("avg_pitch", 0) or 0),
            tempo=float(data.get("tempo", 0) or 0), tempo_conf=float(data.get("tempo_confidence", 0) or 0), emotion=emotion,
            duration=float(data.get("wav_len", 0) or 0))
        return caption, caption + f"Musical: {prompt}"

    def __getitem__(self, index):
        data = self.items[index]
        ori_caption, caption = self._caption(data)
        up = self.unit_upsample_rate
        try:
            spec = np.load(data["mel_path"])                                       # [80, T]
        except Exception:
            print(f"corrupted: {data['mel_path']}")
            spec = np.ones((self.mel_num, self.min_batch_len), dtype=np.float32) * self.pad_value
        org_spec_len = spec_len = spec.shape[1]
        start = 0
        acoustic = np.load(data["vocal_mel_path"])[:20, :]
        midi = np.expand_dims(np.asarray(data["midi"]), axis=0)                    # [1, T]
        beats = np.expand_dims(np.asarray(data["beats"]), axis=0)
        if abs(math.ceil(acoustic.shape[1] * up) - org_spec_len) > 5:              # some bad mel could exist (:282)
            print(f"corrupted: {data['vocal_mel_path']}")
            n = math.ceil(spec_len / up)
            acoustic = np.ones((20, n), dtype=np.float32) * self.unit_pad_value
            midi = np.ones((1, n), dtype=np.float32) * 128
            beats = np.ones((1, n), dtype=np.float32) * 2
        acoustic_len = acoustic.shape[1] + 75                                      # :287
        if math.ceil(acoustic_len * up) > org_spec_len:
            start = round(start * up)
            start = max(min(start, acoustic_len - math.ceil(org_spec_len / up) - 1), 0)
            acoustic_len = math.ceil(org_spec_len / up)
            acoustic = acoustic[:, start:start + acoustic_len]
            midi = midi[:, start:start + acoustic_len]
            beats = beats[:, start:start + acoustic_len]
        acoustic_len = int(math.ceil(acoustic_len / self.unit_frames_multiple) * self.unit_frames_multiple)
        acoustic = pad_or_cut_xd(torch.FloatTensor(np.asarray(acoustic, dtype=np.float32)), acoustic_len, dim=1, pad_value=self.pad_value)
        midi = pad_or_cut_xd(torch.FloatTensor(np.asarray(midi, dtype=np.float32)), acoustic_len, dim=1, pad_value=0)
        beats = pad_or_cut_xd(torch.FloatTensor(np.asarray(beats, dtype=np.float32)), acoustic_len, dim=1, pad_value=0)
        spec_len = math.ceil(acoustic_len * up)
        spec = pad_or_cut_xd(torch.FloatTensor(np.asarray(spec, dtype=np.float32)), spec_len, dim=1, pad_value=self.pad_value)
        assert spec.shape[1] == acoustic.shape[1]
        return {"acoustic": acoustic, "image": spec, "ori_caption": ori_caption, "caption": caption, "name": data["name"], "midi": midi,
                "beats": beats, "audio_path": data.get("audio_path")}

    def collator(self, samples):
        return samples
