"""Second restatement, CMD side: a decoder for EVERY command of a reference-written CMD stream, in Python from the .rs files alone,
on the primitives of ref_restatement.py (Cdf, AnsDecoder).  Test infrastructure; used on the one compressed vector the reference tree
holds (wasm/wasm.html:98-107) next to the C oracle's orc_cmd_stream_walk.  Priors are created on first use (every row starts as the
default CDF, ffi/alloc_util.rs:77-79), keyed by (struct, prior type, index tuple) as codec/priors.rs names them."""
import ref_restatement as rr

MED, FAST, PLANE, ROCKET, SLOW, MUD = (0x30, 0x4000), (0x60, 0x4000), (0x80, 0x4000), (0x180, 0x4000), (0x20, 0x1000), (0x10, 0x2000)
DICT_BITS = [0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5]       # codec/dict.rs:37-40


def round_up_mod_4(v):                                       # codec/interface.rs:180-182
    return (((v - 1) & 0xff) | 3) + 1


def u8_to_speed(d):                                          # probability/interface.rs:577-585
    if d < 8:
        return 0
    lg = (d >> 3) - 1
    return (1 << lg) | (((d & 7) << lg) >> 3)


class CmdWalk:
    def __init__(self, cmd, example_build):
        """example_build: the PredictionMode prior rows of the build that wrote wasm.html's example"""
        self.d = rr.AnsDecoder(bytes(cmd) + b"\0" * 64)
        self.n_bytes = len(cmd)
        self.example_build = example_build
        self.priors = {}
        self.nibbles = 0
        self.last_4_states = 3 << 4                          # codec/interface.rs:375
        self.last_llen, self.last_clen, self.last_dlen = 1, 1, 1
        self.distance_lru = [4, 11, 15, 16]                  # :396
        self.btype_lru = [[0, 1], [0, 1], [0, 1]]
        self.btype_max = [0, 0, 0]
        self.distance_context_map = [i & 3 for i in range(1024)]
        self.cmap_lru = list(range(13))
        self.commands = []
        self.pm = None

    def get(self, key, speed):
        cdf = self.priors.setdefault(key, rr.Cdf())
        sym, _, _ = self.d.get_nibble(cdf)
        cdf.blend(sym, speed)
        self.nibbles += 1
        return sym

    # PredictionModePriors, codec/priors.rs:125-133: offsets of the listed types; an unlisted type falls through to the last entry
    def pm_prior(self, typ, index):
        listed = {"Only": 0, "LiteralSpeed": 1, "FirstNibble": 2, "SecondNibble": 4, "Mnemonic": 6, "PriorMixingValue": 10, "ContextMapSpeedPalette": 27}
        if self.example_build and typ == "Mnemonic":         # that build: row 27 for the literal AND the distance map
            return ("pm", 27)
        return ("pm", listed[typ] + index if typ in listed else 27 + index)

    def prediction_mode(self):                               # context_map.rs:105-428, decoder
        self.cmap_lru = list(range(13))
        self.distance_context_map = [i & 3 for i in range(1024)]
        mode = self.get(self.pm_prior("Only", 0), MED)
        mix = self.get(self.pm_prior("DynamicContextMixingSpeed", 0), MED)
        self.get(self.pm_prior("PriorDepth", 0), FAST)
        f8 = [[0, 0] for _ in range(4)]
        for index in range(16):
            nib = self.get(self.pm_prior("ContextMapSpeedPalette", index & 3), FAST)
            f8[index >> 2][(index & 3) >> 1] |= nib << 3 if (index & 1) == 0 else nib
        maps = [[], []]
        for typ in (0, 1):
            index = 0
            while True:
                mn = self.get(self.pm_prior("Mnemonic", typ), MED)
                if mn == 14:
                    if typ == 0:
                        self.cmap_lru = list(range(13))
                    break
                if mn == 15:
                    val = (self.get(self.pm_prior("FirstNibble", typ), MED) << 4) | self.get(self.pm_prior("SecondNibble", typ), MED)
                else:
                    val = (max(self.cmap_lru) + 1) & 0xff if mn == 13 else self.cmap_lru[mn]
                lru = self.cmap_lru                           # obs_context_map_for_lru, codec/interface.rs:439-467
                if val in lru:
                    lru.remove(val)
                else:
                    lru.pop()
                lru.insert(0, val)
                if typ == 1:
                    self.distance_context_map[index] = val
                maps[typ].append(val)
                index += 1
        mixing = []
        for index in range(8192):
            prior = mixing[index - 256] & 0xf if index >= 256 and not self.example_build else 16
            mixing.append(self.get(self.pm_prior("PriorMixingValue", prior), PLANE))
        self.pm = {"mode": mode, "mixing_math": mix & 3, "speeds": [(u8_to_speed(a), u8_to_speed(b)) for a, b in f8],
                   "literal_context_map": maps[0], "distance_context_map": maps[1], "mixing_values": mixing}

    def block_switch(self, idx):                             # block_type.rs:31-194
        v = self.get(("bt", "Mnemonic", idx), SLOW)
        if v == 0:
            b = self.btype_lru[idx][1]
        elif v == 1:
            b = (self.btype_max[idx] + 1) & 0xff
        elif v == 15:
            first = self.get(("bt", "FirstNibble", idx), SLOW)
            b = (self.get(("bt", "SecondNibble", idx), SLOW) << 4) | first
        else:
            b = v - 2
        stride = self.get(("bt", "StrideNibble", 0), SLOW) if idx == 0 else 0
        self.last_4_states >>= 2
        self.btype_lru[idx] = [b, self.btype_lru[idx][0]]
        self.btype_max[idx] = max(self.btype_max[idx], b)
        return b, stride

    def literal_length(self):                                # literal.rs:565-661
        ctype = self.btype_lru[1][0]
        sc = self.get(("ll", "CountSmall", ctype, 0), MED)
        assert sc != 15, "high-entropy literal"
        if sc != 14:
            self.last_llen = sc + 1
            return sc + 1
        beg = self.get(("ll", "SizeBegNib", ctype), MUD)
        if beg <= 1:
            return 15 + beg                                  # last_llen is left as it was on this exit
        if beg == 15:
            last = self.get(("ll", "SizeLastNib", ctype), MUD)
            rem, dec = round_up_mod_4(last + 14), 1 << (last + 14)
        else:
            rem, dec = round_up_mod_4(beg - 1), 1 << (beg - 1)
        while rem:
            rem -= 4
            dec |= self.get(("ll", "SizeMantissaNib", ctype), MUD) << rem
        self.last_llen = dec + 15
        return dec + 15

    def distance_prior(self, copy_len):                      # get_distance_prior, codec/interface.rs:426-430
        return self.distance_context_map[self.btype_lru[2][0] * 4 + min(max(copy_len, 2) - 2, 3)]

    def copy(self):                                          # copy.rs:49-290
        ctype = self.btype_lru[1][0]
        index = ((self.last_4_states >> 4) & 3) + 4 * min(self.last_llen - 1, 3)
        sc = self.get(("cp", "CountSmall", ctype, index), MUD)
        if sc == 15:
            beg = self.get(("cp", "CountBegNib", ctype, 0), FAST)
            if beg == 15:
                last = self.get(("cp", "CountLastNib", ctype, 0), FAST)
                self.last_clen, rem, dec = last + 19, round_up_mod_4(last + 18), 1 << (last + 18)
            else:
                self.last_clen, rem, dec = beg + 4, round_up_mod_4(beg + 3), 1 << (beg + 3)
            done = 0
            while rem:
                rem -= 4
                dec |= self.get(("cp", "CountMantissaNib", ctype, (self.last_clen % 4) + 1 if done == 0 else 0), SLOW) << rem
                done += 4
            num_bytes = dec
        else:
            num_bytes = sc
            self.last_clen = num_bytes.bit_length()
        prior = self.distance_prior(num_bytes)
        mn = self.get(("cp", "DistanceMnemonic", prior, int(self.last_llen < 8)), SLOW)
        if mn != 15:                                         # get_distance_from_mnemonic_code, codec/interface.rs:979-1009
            if mn < 4:
                dist = self.distance_lru[mn]
            else:
                us = mn >> 2
                dist = self.distance_lru[(mn & 2) >> 1] + us - (((-(mn & 1)) & us) << 1)
            assert dist > 0
            self.last_dlen = dist.bit_length()
        else:
            beg = self.get(("cp", "DistanceBegNib", prior, num_bytes.bit_length() >> 2), SLOW)
            if beg == 15:
                dist = (self.distance_lru[1] - 3) & 0xffffffff
                self.last_dlen = dist.bit_length()
            else:
                if beg == 14:
                    last = self.get(("cp", "DistanceLastNib", prior, 0), ROCKET)
                    self.last_dlen, rem, dec = last + 15, round_up_mod_4(last + 14), 1 << (last + 14)
                else:
                    self.last_dlen, rem, dec = beg + 1, (round_up_mod_4(beg) if beg else 0), 1 << beg
                done = 0
                for sr2 in reversed(range((rem + 3) >> 2)):
                    index = ((self.last_dlen & 3) + 1) if done == 0 else 0
                    inc = 0x4 << ((index & 6) << ((index & 2) >> 1))
                    dec |= self.get(("cp", "DistanceMantissaNib", prior, index), (inc, 0x4000)) << (sr2 << 2)
                    done += 4
                dist = dec
        lru = self.distance_lru                              # obs_distance, codec/interface.rs:509-527
        if dist == lru[1]:
            self.distance_lru = [dist, lru[0], lru[2], lru[3]]
        elif dist == lru[2]:
            self.distance_lru = [dist, lru[0], lru[1], lru[3]]
        elif dist != lru[0]:
            self.distance_lru = [dist, lru[0], lru[1], lru[2]]
        return dist, num_bytes

    def dict(self):                                          # dict.rs:36-190 (nibbles; the word itself needs brotli's dictionary)
        ctype = self.btype_lru[1][0]
        beg = self.get(("dc", "SizeBegNib", ctype), MUD)
        ws = self.get(("dc", "SizeLastNib", ctype), MUD) + 19 if beg == 15 else beg + 4
        rem, word_id, done = round_up_mod_4(DICT_BITS[ws]), 0, 0
        while rem:
            rem -= 4
            word_id |= self.get(("dc", "Index", self.distance_prior(ws), (DICT_BITS[ws] % 4) + 1 if done == 0 else 0), MUD) << rem
            done += 4
        hi = self.get(("dc", "Transform", 0, ws >> 1), FAST)
        lo = self.get(("dc", "Transform", 1, hi), FAST)
        return ws, word_id, (hi << 4) | lo

    def run(self):
        """-> list of (command nibble, payload); stops at the end marker"""
        while True:
            code = self.get(("cc", "FullSelection", self.last_4_states >> 4, 0), ROCKET)      # codec/mod.rs:662-688
            if code == 15:
                self.commands.append((15, None))
                return self.commands
            if code == 7:
                self.prediction_mode()
                self.commands.append((7, None))
            elif code in (4, 5, 6):
                self.commands.append((code, self.block_switch(code - 4)))
            elif code in (1, 2, 3):
                self.last_4_states = (self.last_4_states >> 2) | {1: 64, 2: 192, 3: 128}[code]  # obs_copy / dict / literal_state
                self.commands.append((code, {1: self.copy, 2: self.dict, 3: self.literal_length}[code]()))
            else:
                raise ValueError("command nibble %d" % code)
