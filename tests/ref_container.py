"""A second restatement of the reference's literal-only internal compressor as a WHOLE -- ring-buffer command emission, CMD coder,
the two-stream Mux, header and CRC trailer, under a caller that hands fixed-size output buffers to every call -- in pure Python,
written from the Rust sources alone (not from oracle/stream.c or divans_amd/csrc/host_stream.cpp), on top of the literal coder of
tests/ref_restatement.py.  VERDICT r02 item 6: a misreading of the container would have to be made twice to go unnoticed.
Test infrastructure only; tens of microseconds per nibble.

Sources (relative to the reference tree):
  src/raw_to_cmd/mod.rs:55-181        RawToCmdState::stream / flush (when the 2^window ring emits PredictionMode / Literal commands)
  src/divans_compressor.rs:276-426    DivansCompressor::encode / flush (header, freeze-dried commands, the call structure)
  src/codec/mod.rs:409-560,620-1020   DivansCodec::flush / encode_or_decode_one_command (command type nibble, end marker, trailer)
  src/codec/context_map.rs:105-428    PredictionModeState::encode_or_decode
  src/codec/literal.rs:261-394,496-661  literal length nibbles, code_nibble_array's drains
  src/codec/interface.rs:341-545,683-712,868-918  CrossCommandBookKeeping, drain_or_fill_static_buffer
  src/codec/priors.rs, src/priors.rs:190-237  prior tables (a prior type a table does not list falls through to the LAST listed type)
  src/ans.rs:59-131,253-400           ByteStack / ANSEncoder (the 65 536-symbol chunk, what a drain pops)
  src/mux.rs:36-75,166-342,445-561    Mux: chunk_size / get_code, prep_push_for_n_bytes, serialize, serialize_close
  src/codec/crc32.rs                  CRC-32C
brotli::enc::interface (crate brotli ~3.1, not in the reference tree) supplies the PredictionModeContextMap layout; the internal
compressor only ever builds the one raw_to_cmd/mod.rs:115-140 builds: prediction mode 0 (LSB6), literal context map i & 0x3f for
64 entries, distance context map i & 3 for 4, every mixing value 4, all speed bytes 0.

A call that runs out of output returns NEEDS_MORE_OUTPUT and is re-entered with a fresh buffer; re-entry replays the frozen commands
(divans_compressor.rs:189-207) and resumes the codec's state machine where it stopped, so the model below simply continues with a
new buffer (`_Caller.fresh`) at every point where the reference returns to its caller.
"""
import ref_restatement as rr

MED, MUD, FAST, PLANE, ROCKET = (0x30, 0x4000), (0x10, 0x2000), (0x60, 0x4000), (0x80, 0x4000), (0x180, 0x4000)   # probability/interface.rs:321-328
MAX_HEADER_SIZE = 3
MAX_FLUSH_VARIANCE = 131073
NUM_MIXING_VALUES = 8192


def crc32c(data, crc=0):                                     # codec/crc32.rs (Castagnoli, reflected)
    crc ^= 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def speed_to_u8(v):                                          # probability/interface.rs:566-575
    length = v.bit_length()
    mantissa = (((v - (1 << (length - 1))) << 3) >> (length - 1)) if v else 0
    return ((length << 3) | mantissa) & 0xFF


def u8_to_speed(d):                                          # :577-585
    if d < 8:
        return 0
    log_val = (d >> 3) - 1
    return rr.i16((1 << log_val) | (((d & 7) << log_val) >> 3))


class _Caller:
    """The application around the compressor (c/example.c:26-60): every call brings an empty buffer of `cap` bytes."""

    def __init__(self, cap):
        self.cap = cap; self.cur = bytearray(); self.done = bytearray()

    def room(self):
        return self.cap - len(self.cur)

    def fresh(self):
        self.done += self.cur; self.cur = bytearray()


class _Queue(rr.AnsEncoder):
    """ANSEncoder with its output ByteStack `q` (ans.rs:253-400): flush_chunk leaves the chunk's bytes there, a drain pops them."""

    def has_data(self):
        return len(self.out) != 0

    def pop(self, n):
        got = bytes(self.out[:n]); del self.out[:n]
        return got

    def put_nibble(self, sym, cdf):                          # put_nibble_internal :277-286
        start, freq = cdf.sym_to_start_and_freq(sym)
        self.put_start_freq(start, freq)


class Mux:                                                   # src/mux.rs
    def __init__(self):
        self.buf = [bytearray(), bytearray()]; self.start = [0, 0]; self.end = [0, 0]
        self.cur_stream_bytes_avail = 0; self.cur_stream = 0
        self.last_flush = [0, 0]; self.bytes_flushed = 0; self.eof = 0      # 0 Running, 1 EofStart, 2 EofMid, 3 EofDone

    @staticmethod
    def chunk_size(last_flushed, lagging):                   # :36-47
        if lagging:
            return 16
        if last_flushed <= 1024:
            return 4096
        if last_flushed <= 65536:
            return 16384
        return 65536

    @staticmethod
    def get_code(sid, n, lagging):                           # :54-75 -> (header bytes, payload bytes)
        if (not lagging) or n == 4096 or n == 16384 or n >= 65536:
            if n < 4096:
                return Mux.get_code(sid, n, True)
            if n < 16384:
                return bytes([sid | (1 << 4)]), 4096
            if n < 65536:
                return bytes([sid | (2 << 4)]), 16384
            return bytes([sid | (3 << 4)]), 65536
        assert n < 65536
        return bytes([sid, (n - 1) & 0xFF, ((n - 1) >> 8) & 0xFF]), n

    def prep_push(self, sid, n):                             # prep_push_for_n_bytes :246-288
        buf = self.buf[sid]; wc = self.end[sid]; rc = self.start[sid]
        if len(buf) - wc >= n:
            return
        if len(buf) >= (wc - rc) + n + MAX_HEADER_SIZE and (rc == wc or (rc >= 16384 and rc > wc - rc + MAX_HEADER_SIZE)):
            amount = wc - rc
            buf[MAX_HEADER_SIZE:MAX_HEADER_SIZE + amount] = buf[rc:rc + amount]
            self.end[sid] = MAX_HEADER_SIZE + amount; self.start[sid] = MAX_HEADER_SIZE
            return
        desired = MAX_HEADER_SIZE + n + (wc - rc)
        log_desired = desired.bit_length() + 1
        new = bytearray(1 << max(log_desired, 9))
        new[MAX_HEADER_SIZE:MAX_HEADER_SIZE + (wc - rc)] = buf[rc:wc]
        self.end[sid] = MAX_HEADER_SIZE + wc - rc; self.start[sid] = MAX_HEADER_SIZE
        self.buf[sid] = new

    def write_buffer(self):                                  # :171-191
        for sid in (0, 1):
            self.prep_push(sid, 16)

    def push_from(self, sid, coder):
        """drain_or_fill_internal_buffer_unchecked into cur_output[sid]: the coder pops into what is left of the stream's buffer"""
        got = coder.pop(len(self.buf[sid]) - self.end[sid])
        self.buf[sid][self.end[sid]:self.end[sid] + len(got)] = got
        self.end[sid] += len(got)

    def _leftover(self, out, cap):                           # serialize_leftover :290-297
        n = min(self.cur_stream_bytes_avail, cap)
        s = self.cur_stream
        out += self.buf[s][self.start[s]:self.start[s] + n]
        self.start[s] += n; self.cur_stream_bytes_avail -= n
        return n

    def _stream(self, sid, out, cap, lagging):               # serialize_stream_id :298-342; cap = room left in the caller's buffer
        hdr, n = Mux.get_code(sid, self.end[sid] - self.start[sid], lagging)
        self.bytes_flushed += n
        assert self.start[sid] >= MAX_HEADER_SIZE
        n += len(hdr)
        self.start[sid] -= len(hdr)
        self.buf[sid][self.start[sid]:self.start[sid] + len(hdr)] = hdr
        self.last_flush[sid] = self.bytes_flushed
        w = min(n, cap)
        out += self.buf[sid][self.start[sid]:self.start[sid] + w]
        self.start[sid] += w
        if self.start[sid] == self.end[sid]:
            self.start[sid] = MAX_HEADER_SIZE; self.end[sid] = MAX_HEADER_SIZE
        if w != n:
            self.cur_stream_bytes_avail = n - w; self.cur_stream = sid
        return w

    def serialize(self, cap):                                # :445-477
        out = bytearray()
        if self.cur_stream_bytes_avail:
            self._leftover(out, cap)
        while len(out) < cap:
            flushed_any = False
            last_flush = min(self.last_flush); max_flush = max(self.last_flush)
            for sid in (0, 1):
                lagging = max_flush > MAX_FLUSH_VARIANCE + self.last_flush[sid]
                if (self.end[sid] - self.start[sid] >= Mux.chunk_size(self.last_flush[sid], lagging)
                        and self.last_flush[sid] <= last_flush + MAX_FLUSH_VARIANCE):
                    flushed_any = True
                    self._stream(sid, out, cap - len(out), lagging)
                    if self.cur_stream_bytes_avail:
                        break
            if not flushed_any:
                break
        return bytes(out)

    def _flush_internal(self, out, cap):                     # :520-560
        if self.cur_stream_bytes_avail:
            self._leftover(out, cap)
        while len(out) < cap:
            flushed_any = False
            last_flush = None
            for sid in (0, 1):
                has = self.start[sid] != self.end[sid]
                if (last_flush is None and has) or (last_flush is not None and self.last_flush[sid] < last_flush and has):
                    last_flush = self.last_flush[sid]
            for sid in (0, 1):
                if last_flush is None or self.last_flush[sid] <= last_flush + MAX_FLUSH_VARIANCE:
                    before = len(out)
                    if self.start[sid] != self.end[sid]:
                        self._stream(sid, out, cap - len(out), True)
                    if len(out) != before:
                        flushed_any = True
                    if self.cur_stream_bytes_avail:
                        break
            if not flushed_any:
                break

    def serialize_close(self, cap):                          # :478-519
        if self.eof == 3:
            return b""
        out = bytearray()
        self._flush_internal(out, cap)
        while self.eof < 3 and len(out) < cap:               # EOF_MARKER, one byte per state
            out.append((0xFF, 0xFE, 0xFF)[self.eof]); self.eof += 1
        return bytes(out)

    def is_eof(self):                                        # :216-226
        return self.start == self.end and self.eof == 3


class _Priors:
    """define_prior_struct!: one default-initialised CDF per (type, index); a type the table does not list lands in the LAST listed
    type's region (src/priors.rs:226-237: the final macro arm does not test the type)."""

    def __init__(self, listed):
        self.listed = listed; self.t = {}

    def get(self, typ, index):
        if typ not in self.listed:
            typ = self.listed[-1]
        return self.t.setdefault((typ, index), rr.Cdf())


class Compressor:
    """DivansCompressor<ANSEncoder> of the C FFI with use_brotli = UseInternalCommandSelection (src/ffi/compressor.rs).

    The coding of a command is a generator that yields wherever the reference returns NEEDS_MORE_OUTPUT to its caller; a command
    interrupted that way is what divans_compressor.rs:189-207 freeze-dries, and the next call (whichever it is) resumes it first."""

    def __init__(self, window_size=22, dynamic_context_mixing=0, prior_depth=0, use_context_map=False, force_stride=0,
                 literal_adaptation=None, call_buffer_size=65536, stale_tail=False):
        self.window = min(24, max(10, window_size))          # divans_compressor.rs:89
        self.ring = bytearray(1 << self.window)
        self.dec = 0; self.outi = 0; self.has_produced_header = False       # RawToCmdState
        # raw_to_cmd/mod.rs:70-72 resets the write index after the few bytes a lap first writes at the END of the ring even when the
        # input ran out inside them, and flush (:143-150) then emits that whole span, stale bytes of the previous lap included: the
        # reference's own container then decodes to more than was put in.  stale_tail=True is that behaviour, literally; the
        # default emits the fresh bytes only -- the one deliberate deviation of the oracle and of the product (oracle/stream.c:646-650)
        self.stale_tail = stale_tail; self.tail_fresh = 0
        self.caller = _Caller(call_buffer_size)
        self.header_progress = 0
        self.cmd = _Queue(); self.lit = _Queue(); self.mux = Mux()
        dcm = dynamic_context_mixing
        if force_stride != 0 and dcm == 0 and use_context_map:               # codec/interface.rs:361-366 (0 = PriorDisabled)
            dcm = 1
        self.desired_context_mixing = dcm; self.desired_prior_depth = prior_depth
        self.desired_adapt = literal_adaptation; self.do_context_map = use_context_map
        self.last_4_states = 3 << 4                          # :373
        self.cmap_lru = [0] * 13
        self.cc = _Priors(["FullSelection", "EndIndicator"])
        self.litlen = _Priors(["CountSmall", "SizeBegNib", "SizeLastNib", "SizeMantissaNib"])
        self.pred = _Priors(["Only", "LiteralSpeed", "FirstNibble", "SecondNibble", "Mnemonic", "PriorMixingValue", "ContextMapSpeedPalette"])
        self.lc = None                                       # literal coder: configured by the PredictionMode command
        self.pending = None                                  # the freeze-dried commands' coroutine
        self.finishing = None                                # DivansCodec::flush's

    # ---- output plumbing ------------------------------------------------------------------------------------------
    def _drain(self, coder, sid):                            # drain_or_fill_static_buffer, codec/interface.rs:868-896 (can_linearize)
        while coder.has_data():
            self.caller.cur += self.mux.serialize(self.caller.room())
            self.mux.write_buffer()
            self.mux.push_from(sid, coder)
            if coder.has_data() and self.caller.room() == 0:
                return False                                 # NEEDS_MORE_OUTPUT
        return True

    def _need(self, coder, sid):                             # a drain whose failure goes back to the caller; re-entry repeats it
        while not self._drain(coder, sid):
            yield

    def _put(self, nib, cdf, speed):                         # get_or_put_nibble + blend on the CMD coder
        self.cmd.put_nibble(nib, cdf)
        cdf.blend(nib, speed)

    # ---- commands -------------------------------------------------------------------------------------------------
    def _command_type(self, code):                           # codec/mod.rs:652-676
        yield from self._need(self.cmd, 0)
        self._put(code, self.cc.get("FullSelection", (self.last_4_states >> 4, 0)), ROCKET)

    def _prediction_mode(self):                              # context_map.rs:105-428 for the command raw_to_cmd/mod.rs:115-140 builds
        yield from self._command_type(7)
        desired = [MUD] * 4                                  # default_literal_speed; the command's speed bytes are all zero
        if self.desired_adapt is not None:
            desired = list(self.desired_adapt)
        yield from self._need(self.cmd, 0)
        self.cmap_lru = list(range(13))                      # reset_context_map_lru
        self._put(0, self.pred.get("Only", (0,)), MED)       # literal_prediction_mode: byte 0 of a zeroed buffer = LSB6
        yield from self._need(self.cmd, 0)
        nib = self.desired_context_mixing | (0 << 3)
        self._put(nib, self.pred.get("DynamicContextMixingSpeed", (0,)), MED)
        mixing_math = nib & 3
        combine = nib != 0
        yield from self._need(self.cmd, 0)
        self._put(self.desired_prior_depth, self.pred.get("PriorDepth", (0,)), FAST)
        f8 = [[0, 0] for _ in range(4)]
        for index in range(16):
            cur = (speed_to_u8(desired[index >> 2][0]), speed_to_u8(desired[index >> 2][1]))
            typ = index & 3
            nib = [(cur[0] & 0x7F) >> 3, (cur[0] & 0x7F) & 7, (cur[1] & 0x7F) >> 3, (cur[1] & 0x7F) & 7][typ]
            yield from self._need(self.cmd, 0)
            self._put(nib, self.pred.get("ContextMapSpeedPalette", (typ,)), FAST)
            f8[index >> 2][typ >> 1] |= (nib << 3) if typ in (0, 2) else nib
        # set_stride_context_speed / set_context_map_speed store speed_to_u8(u8_to_speed(f8)); LiteralBookKeeping reads them back
        # with from_f8_tuple (codec/interface.rs:304-309)
        speeds = [(u8_to_speed(speed_to_u8(u8_to_speed(a))), u8_to_speed(speed_to_u8(u8_to_speed(b)))) for a, b in f8]
        cmap = [0] * 16384                                   # the PredictionModeState's own buffer: fresh from the allocator
        for kind, src in ((0, [i & 0x3F for i in range(64)]), (1, [i & 3 for i in range(4)])):
            if not self.do_context_map:
                src = []
            index = 0
            while True:
                yield from self._need(self.cmd, 0)
                if index >= len(src):
                    mnemonic = 14
                else:
                    target = src[index]
                    mnemonic = 15
                    for i, v in enumerate(self.cmap_lru):
                        if v == target:
                            mnemonic = i
                    if target == (max(self.cmap_lru) + 1) & 0xFF:
                        mnemonic = 13
                self._put(mnemonic, self.pred.get("Mnemonic", (kind,)), MED)
                if mnemonic == 14:
                    if kind == 0:
                        self.cmap_lru = list(range(13))
                    break
                if mnemonic == 15:
                    val = src[index]
                    yield from self._need(self.cmd, 0)
                    self._put(val >> 4, self.pred.get("FirstNibble", (kind,)), MED)
                    yield from self._need(self.cmd, 0)
                    self._put(val & 0xF, self.pred.get("SecondNibble", (kind,)), MED)
                else:
                    val = (max(self.cmap_lru) + 1) & 0xFF if mnemonic == 13 else self.cmap_lru[mnemonic]
                self._obs_lru(val)
                if kind == 0:
                    cmap[index] = val
                index += 1
        mixing = [0] * NUM_MIXING_VALUES
        for index in range(NUM_MIXING_VALUES):
            nib = 4 if not self.do_context_map else (0 if not combine else 4)
            prior = (mixing[index - 256] & 0xF) if index >= 256 else 16
            yield from self._need(self.cmd, 0)
            self._put(nib, self.pred.get("PriorMixingValue", (prior,)), PLANE)
            mixing[index] = nib
        # obs_prediction_mode_context_map, codec/interface.rs:293-321
        assert self.lc is None                               # the internal compressor sends exactly one PredictionMode
        self.lc = rr.LiteralCoder(cmap, mixing, 0, 0, mixing_math, speeds)

    def _obs_lru(self, val):                                 # obs_context_map_for_lru, codec/interface.rs:439-467
        lru = self.cmap_lru
        if val in lru:
            i = lru.index(val)
            if i != 0:
                lru[1:i + 1] = lru[:i]
        else:
            lru[1:] = lru[:-1]
        lru[0] = val

    def _literal(self, data):                                # codec/literal.rs:496-661 then :261-394
        yield from self._command_type(3)
        self.last_4_states = (self.last_4_states >> 2) | 128                 # obs_literal_state
        n = len(data)
        serialized = (n - 15) & 0xFFFFFFFF
        lllen = serialized.bit_length()
        yield from self._need(self.cmd, 0)                   # Begin -> LiteralCountSmall
        yield from self._need(self.cmd, 0)
        shortcut = min(14, (n - 1) & 0xFFFFFFFF)
        self._put(shortcut, self.litlen.get("CountSmall", (0, 0)), MED)
        if shortcut == 14:
            beg = min(15, lllen)
            yield from self._need(self.cmd, 0)
            self._put(beg, self.litlen.get("SizeBegNib", (0,)), MUD)
            rem = 0
            if beg == 15:
                last = (lllen - 15) & 0xFF
                yield from self._need(self.cmd, 0)
                self._put(last, self.litlen.get("SizeLastNib", (0,)), MUD)
                rem, so_far = ((last + 14 - 1) | 3) + 1, 1 << (last + 14)
            elif beg > 1:
                rem, so_far = ((beg - 1 - 1) | 3) + 1, 1 << (beg - 1)
            while rem:
                nxt = rem - 4
                nib = ((serialized ^ so_far) >> nxt) & 0xFF
                yield from self._need(self.cmd, 0)
                self._put(nib, self.litlen.get("SizeMantissaNib", (0,)), MUD)
                so_far |= nib << nxt
                rem = nxt
        lc = self.lc
        yield from self._need(self.cmd, 0)
        yield from self._need(self.lit, 1)                   # encode_or_decode_content_bytes: the LIT coder is drained before a nibble
        for k in range(n):                                   # code_nibble_array (NibbleArrayLowBuffer: the encoder's demuxer has no data)
            byte = data[k]
            prev = (lc.last_8 >> 0x38) & 0xFF; pp = (lc.last_8 >> 0x30) & 0xFF
            ctx = lc.cmap[(lc.lut0[prev] | lc.lut1[pp]) + (lc.btype << 6)]
            stride_bytes = lc.last_8
            h, prob = lc._code_nibble(True, byte >> 4, ctx, prev, stride_bytes, 0, self.lit, None)
            ok = self._drain(self.lit, 1)
            if prob is not None:
                prob.blend(h, lc.speeds[0])
            if not ok:                                       # fallback_byte_encode: back to the caller, the low nibble after re-entry
                yield
                yield from self._need(self.cmd, 0)
                yield from self._need(self.lit, 1)
            lo, prob = lc._code_nibble(False, byte & 0xF, ctx, prev, stride_bytes, h, self.lit, None)
            lc.last_8 = (lc.last_8 >> 8) | ((lo | (h << 4)) << 0x38)
            if prob is not None:
                prob.blend(lo, lc.speeds[0])
            ok = self._drain(self.lit, 1)
            if not ok and k + 1 != n:                        # after the LAST byte the status is dropped (literal.rs:383-386): the command
                yield                                        # completes and the LIT coder keeps its bytes until the next LIT drain
                yield from self._need(self.cmd, 0)
                yield from self._need(self.lit, 1)

    def _run(self, cmds):
        for kind, data in cmds:
            if kind == "pm":
                yield from self._prediction_mode()
            else:
                yield from self._literal(data)

    # ---- RawToCmdState ---------------------------------------------------------------------------------------------
    def _ring_full(self):                                    # raw_to_cmd/mod.rs:52-54
        return self.dec == len(self.ring) or self.dec + 1 == self.outi

    def _assembler_flush(self):                              # :105-181 -> the commands (ring bytes copied out)
        cmds = []
        if not self.has_produced_header:
            self.has_produced_header = True
            cmds.append(("pm", None))
        if self.dec < self.outi:
            tail = bytes(self.ring[self.outi:] if self.stale_tail else self.ring[self.outi:self.outi + self.tail_fresh])
            self.tail_fresh = 0
            if tail:
                cmds.append(("lit", tail))
            if self.dec == len(self.ring):
                self.dec = 0
            self.outi = 0
        if self.dec != self.outi:
            cmds.append(("lit", bytes(self.ring[self.outi:self.dec])))
            self.outi = self.dec
        return cmds

    def _assembler_stream(self, data, pos):                  # :55-104 -> (commands, new position, all input taken)
        if self.dec >= self.outi:
            m = min(len(self.ring) - self.dec, len(data) - pos)
            self.ring[self.dec:self.dec + m] = data[pos:pos + m]
            pos += m; self.dec += m
            if self.outi != 0:
                self.tail_fresh = self.dec - self.outi
                self.dec = 0
        if self.dec < self.outi:
            m = min(self.outi - 1 - self.dec, len(data) - pos)
            self.ring[self.dec:self.dec + m] = data[pos:pos + m]
            pos += m; self.dec += m
        cmds = []
        if self._ring_full():
            cmds = self._assembler_flush()
            if pos != len(data):
                return cmds, pos, False
        assert pos == len(data)
        return cmds, pos, True

    # ---- the calls -------------------------------------------------------------------------------------------------
    def _header(self):                                       # write_header, divans_compressor.rs:147-171 -> complete?
        hdr = bytes([0xFF, 0xE5, 0x8C, 0x9F, 0, self.window]) + bytes(10)
        n = min(self.caller.room(), 16 - self.header_progress)
        self.caller.cur += hdr[self.header_progress:self.header_progress + n]
        self.header_progress += n
        return self.header_progress == 16

    def _resume(self, attr):                                 # -> True when the coroutine ran to its end
        gen = getattr(self, attr)
        if gen is None:
            return True
        try:
            next(gen)
            return False
        except StopIteration:
            setattr(self, attr, None)
            return True

    def encode_call(self, data, pos):                        # DivansCompressor::encode, :276-337 -> new position
        if not self._header() or not self._resume("pending"):
            return pos
        while True:
            cmds, pos, done = self._assembler_stream(data, pos)
            if done and not cmds:
                return pos
            self.pending = self._run(cmds)
            if not self._resume("pending") or done:
                return pos

    def flush_call(self):                                    # DivansCompressor::flush, :362-426 -> finished?
        if not self._header() or not self._resume("pending"):
            return False
        if self.finishing is None:
            cmds = self._assembler_flush()
            if cmds:
                self.pending = self._run(cmds)
                if not self._resume("pending"):
                    return False
            self.finishing = self._finish()
        return self._resume("finishing")

    def _finish(self):                                       # DivansCodec::internal_flush, codec/mod.rs:424-560
        yield from self._command_type(0xF)                   # the end marker through the command-type prior
        while not (self._drain(self.cmd, 0) and self._drain(self.lit, 1)):   # EncodedShutdownNode (both again after a re-entry)
            yield
        self.cmd.flush_chunk(); self.lit.flush_chunk()       # ShutdownCoder(0), (1)
        while not (self._drain(self.cmd, 0) and self._drain(self.lit, 1)):   # CoderBufferDrain
            yield
        while True:                                          # MuxDrain
            if self.caller.room() == 0:
                yield
                continue
            self.caller.cur += self.mux.serialize_close(self.caller.room())
            if self.mux.is_eof():
                break
        crc = crc32c(bytes(self.caller.done) + bytes(self.caller.cur))
        trailer = bytes([crc & 0xFF, (crc >> 8) & 0xFF, (crc >> 16) & 0xFF, (crc >> 24) & 0xFF]) + b"ans~"   # WriteChecksum
        count = 0
        while True:
            n = min(self.caller.room(), 8 - count)
            self.caller.cur += trailer[count:count + n]
            count += n
            if count == 8:
                return
            yield


def compress(data, call_inputs=None, **options):
    """The application of c/example.c:26-60: divans_encode until the input is taken (a fresh buffer for every call), then
    divans_encode_flush until it reports success; `call_inputs` cuts the input into the pieces successive encode loops are given."""
    c = Compressor(**options)
    data = bytes(data)
    if call_inputs is None:
        call_inputs = [len(data)]
    start = 0
    for n in call_inputs:
        piece = data[start:start + n]; start += n
        pos = 0
        while pos < len(piece):
            pos = c.encode_call(piece, pos)
            c.caller.fresh()
    assert start == len(data)
    while True:
        finished = c.flush_call()
        c.caller.fresh()
        if finished:
            return bytes(c.caller.done)
