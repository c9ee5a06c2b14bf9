"""The reference's 24-function dabsdr API driven exactly as src/radiocontrol.cpp drives it
(init -> register callbacks -> dabsdr() -> Request_Tune -> notifications), with a raw-file
style input callback (float(u8-128), blocking fill) — GPU only."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import abracadabra_amd as aa
from oracle import binding as ob

pytestmark = pytest.mark.gpu

NID = dict(SYNC_STATUS=1, TUNE=2, ENSEMBLE_INFO=3, SERVICE_LIST=4, SERVICE_COMPONENT_LIST=5, SERVICE_SELECTION=8,
           USER_APP_LIST=7, PERIODIC=10, RESET=13, ANNOUNCEMENT_SUPPORT=14, ANNOUNCEMENT_SWITCHING=15, PTY=16, TII=17)


class Label(C.Structure):
    _fields_ = [("str", C.c_char * 17), ("charField", C.c_uint16), ("charset", C.c_uint8)]


class Ensemble(C.Structure):
    _fields_ = [("frequency", C.c_uint32), ("ueid", C.c_uint32), ("LTO", C.c_int8), ("intTable", C.c_uint8),
                ("alarm", C.c_uint8), ("label", Label)]


class Periodic(C.Structure):
    _fields_ = [("syncLevel", C.c_int), ("snr10", C.c_int16), ("freqOffset", C.c_int32), ("dateHoursMinutes", C.c_uint32),
                ("secMsec", C.c_uint16), ("fibErrorCntr", C.c_uint16), ("mscCrcOkCntr", C.c_uint8), ("mscCrcErrorCntr", C.c_uint8),
                ("audioServiceBytes", C.c_uint16), ("padBytes", C.c_uint16), ("rsUncorrectableCntr", C.c_uint16),
                ("rsBitErrors", C.c_uint16), ("rsBytes", C.c_uint16)]


class ServiceItem(C.Structure):
    _fields_ = [("sid", C.c_uint32), ("label", Label), ("pty_s", C.c_uint8), ("pty_d", C.c_uint8), ("CAId", C.c_uint8)]


class Ntf(C.Structure):
    _fields_ = [("nid", C.c_int), ("status", C.c_int), ("len", C.c_uint16), ("pData", C.c_void_p)]


class CompItem(C.Structure):                       # dabsdrServiceCompListItem_t (dabsdr.h:193-243)
    class _U(C.Union):
        class _A(C.Structure):
            _fields_ = [("ASCTy", C.c_uint8), ("bitRate", C.c_uint16)]

        class _P(C.Structure):
            _fields_ = [("DSCTy", C.c_uint8), ("SCId", C.c_uint16), ("DGflag", C.c_uint8), ("packetAddress", C.c_int16)]
        _fields_ = [("streamAudio", _A), ("packetData", _P)]
    _fields_ = [("SCIdS", C.c_uint8), ("SubChId", C.c_uint8), ("SubChAddr", C.c_int16), ("SubChSize", C.c_uint16),
                ("protectionLevel", C.c_uint8), ("uepIdx", C.c_uint8), ("ps", C.c_uint8), ("lang", C.c_uint8), ("CAflag", C.c_uint8),
                ("label", Label), ("numUserApps", C.c_uint8), ("TMId", C.c_uint8), ("u", _U)]


class CompList(C.Structure):
    _fields_ = [("SId", C.c_uint32), ("numServiceComponents", C.c_uint8),
                ("getItem", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint8, C.POINTER(CompItem)))]


class UserAppItem(C.Structure):
    _fields_ = [("type", C.c_uint16), ("label", Label), ("dataLen", C.c_uint8), ("data", C.c_uint8 * 23)]


class UserAppList(C.Structure):
    _fields_ = [("SId", C.c_uint32), ("SCIdS", C.c_uint8), ("numUserApps", C.c_uint8),
                ("getItem", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint8, C.POINTER(UserAppItem)))]


class AnnSupport(C.Structure):
    _fields_ = [("SId", C.c_uint32), ("ASu", C.c_uint16), ("numClusterIds", C.c_uint8), ("clusterIds", C.c_uint8 * 7)]


class Asw(C.Structure):
    _fields_ = [("clusterId", C.c_uint8), ("subChId", C.c_uint8), ("ASwFlags", C.c_uint16)]


class PTy(C.Structure):
    _fields_ = [("SId", C.c_uint32), ("s", C.c_uint8), ("d", C.c_uint8)]


class TiiId(C.Structure):
    _fields_ = [("main", C.c_uint8), ("sub", C.c_uint8), ("level", C.c_float)]


class NtfTii(C.Structure):
    _fields_ = [("numIds", C.c_uint8), ("id", TiiId * 24), ("getSpectrumTii", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float)))]


class ServiceList(C.Structure):
    _fields_ = [("numServices", C.c_uint8), ("getItem", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint8, C.POINTER(ServiceItem)))]


def test_periodic_struct_is_32_bytes():
    assert C.sizeof(Periodic) == 32          # the host asserts this length (radiocontrol.cpp:2407)


def test_tune_lock_ensemble_and_service_list():
    L = aa.load_library()
    cfo = 1500.0
    sub = [[0, 0, 3, 64], [48, 1, 4, 32]]
    iq, fib, _ = ob.tx_generate(seed=77, eid=0x1234, n_frames=24, subch=sub, delay=5000, snr_db=25.0, cfo_hz=cfo, tii=(21, 5), extra_figs=True)
    samples = (iq.astype(np.float32) - 128.0)              # what RawFileWorker produces (rawfileinput.cpp:692)
    pos = [0]
    events, lock = [], threading.Lock()
    handle = C.c_void_p()

    @C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_uint16)
    def get_samples(buf, n):
        out = np.ctypeslib.as_array(buf, shape=(2 * n,))
        take = samples[pos[0]:pos[0] + 2 * n]
        out[:len(take)] = take
        out[len(take):] = 0.0                               # flushed FIFO returns zeros (inputdevice.cpp:80-85)
        pos[0] += 2 * n

    @C.CFUNCTYPE(None, C.POINTER(Ntf), C.c_void_p)
    def on_ntf(p, ctx):
        n = p.contents
        rec = dict(nid=n.nid, status=n.status, len=n.len)
        if n.nid == NID["TUNE"]:
            rec["freq"] = C.cast(n.pData, C.POINTER(C.c_uint32)).contents.value
        elif n.nid == NID["SYNC_STATUS"]:
            rec["level"] = C.cast(n.pData, C.POINTER(C.c_int)).contents.value
            rec["snr10"] = C.cast(n.pData + 4, C.POINTER(C.c_int16)).contents.value
        elif n.nid == NID["ENSEMBLE_INFO"]:
            e = C.cast(n.pData, C.POINTER(Ensemble)).contents
            rec.update(ueid=e.ueid, lto=e.LTO, label=e.label.str.decode(), freq=e.frequency)
        elif n.nid == NID["PERIODIC"] and n.pData:
            pr = C.cast(n.pData, C.POINTER(Periodic)).contents
            rec.update(fib_err=pr.fibErrorCntr, foff=pr.freqOffset, level=pr.syncLevel, dhm=pr.dateHoursMinutes, secms=pr.secMsec)
        elif n.nid == NID["SERVICE_COMPONENT_LIST"]:
            cl = C.cast(n.pData, C.POINTER(CompList)).contents
            comps = []
            for i in range(cl.numServiceComponents):
                it = CompItem()
                cl.getItem(handle, i, C.byref(it))
                comps.append(dict(scids=it.SCIdS, subch=it.SubChId, addr=it.SubChAddr, size=it.SubChSize, lang=it.lang, napps=it.numUserApps,
                                  tmid=it.TMId, ascty=it.u.streamAudio.ASCTy, kbps=it.u.streamAudio.bitRate, ps=it.ps))
            rec.update(sid=cl.SId, comps=comps)
        elif n.nid == NID["USER_APP_LIST"]:
            ul = C.cast(n.pData, C.POINTER(UserAppList)).contents
            apps = []
            for i in range(ul.numUserApps):
                it = UserAppItem()
                ul.getItem(handle, i, C.byref(it))
                apps.append((it.type, bytes(it.data[:it.dataLen])))
            rec.update(sid=ul.SId, scids=ul.SCIdS, apps=apps)
        elif n.nid == NID["ANNOUNCEMENT_SUPPORT"]:
            a = C.cast(n.pData, C.POINTER(AnnSupport)).contents
            rec.update(sid=a.SId, asu=a.ASu, clusters=list(a.clusterIds[:a.numClusterIds]))
        elif n.nid == NID["ANNOUNCEMENT_SWITCHING"]:
            a = C.cast(n.pData, C.POINTER(Asw * 8)).contents
            rec.update(asw=[(x.clusterId, x.subChId, x.ASwFlags) for x in a if x.ASwFlags])
        elif n.nid == NID["PTY"]:
            y = C.cast(n.pData, C.POINTER(PTy)).contents
            rec.update(sid=y.SId, s=y.s, d=y.d)
        elif n.nid == NID["TII"]:
            t = C.cast(n.pData, C.POINTER(NtfTii)).contents
            spec = (C.c_float * 384)()                       # the host's buffer size (radiocontrol.h:280)
            t.getSpectrumTii(handle, spec)
            rec.update(ids=[(t.id[i].main, t.id[i].sub) for i in range(t.numIds)], spec=np.array(spec))
        elif n.nid == NID["SERVICE_LIST"]:
            sl = C.cast(n.pData, C.POINTER(ServiceList)).contents
            items = []
            for i in range(sl.numServices):
                it = ServiceItem()
                sl.getItem(handle, i, C.byref(it))           # getter called inside the callback, as the host does
                items.append((it.sid, it.label.str.decode()))
            rec["services"] = items
        with lock:
            events.append(rec)

    L.dabsdrInit.argtypes = [C.POINTER(C.c_void_p)]
    assert L.dabsdrInit(C.byref(handle)) == 0
    for name in ("dabsdrRegisterInputFcn", "dabsdrRegisterDummyInputFcn"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
    L.dabsdrRegisterInputFcn(handle, C.cast(get_samples, C.c_void_p))
    L.dabsdrRegisterDummyInputFcn(handle, C.cast(get_samples, C.c_void_p))
    L.dabsdrRegisterNotificationCb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.dabsdrRegisterNotificationCb(handle, C.cast(on_ntf, C.c_void_p), None)
    for name in ("dabsdr", "dabsdrRequest_GetEnsemble", "dabsdrRequest_GetServiceList", "dabsdrRequest_Exit"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.dabsdrRequest_Tune.argtypes = [C.c_void_p, C.c_uint32]
    L.dabsdrRequest_SetPeriodicNotify.argtypes = [C.c_void_p, C.c_uint8, C.c_uint32]
    L.dabsdr(handle)
    L.dabsdrRequest_SetPeriodicNotify(handle, 1, 0)          # every 2 frames
    L.dabsdrRequest_SetTII.argtypes = [C.c_void_p, C.c_uint8, C.c_int]
    L.dabsdrRequest_SetTII(handle, 1, 1)
    L.dabsdrRequest_Tune(handle, 225648)

    def wait_for(pred, timeout=60.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            with lock:
                hit = [e for e in events if pred(e)]
            if hit:
                return hit
            time.sleep(0.02)
        raise AssertionError(f"timeout; events so far: {events[-10:]}")

    wait_for(lambda e: e["nid"] == NID["TUNE"] and e.get("freq") == 225648)
    wait_for(lambda e: e["nid"] == NID["SYNC_STATUS"] and e.get("level") == 3)       # DABSDR_SYNC_LEVEL_FIC
    fic = wait_for(lambda e: e["nid"] == NID["SYNC_STATUS"] and e.get("level") == 3)[-1]
    assert 200 <= fic["snr10"] <= 300                                                 # signal generated at 25 dB
    per = wait_for(lambda e: e["nid"] == NID["PERIODIC"] and "fib_err" in e and e["level"] == 3)
    assert per[-1]["fib_err"] == 0
    assert abs(per[-1]["foff"] / 10.0 - cfo) < 10.0                                   # Hz, positive = above nominal
    dated = wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e.get("dhm"))[-1]        # FIG 0/10 as the host decodes it
    assert (dated["dhm"] >> 14) & 0x1FFFF == 60587 and (dated["dhm"] >> 6) & 0x1F == 12 and dated["dhm"] & 0x3F == 34   # dabtables.cpp:128-134
    assert dated["secms"] >> 10 == 56 and dated["secms"] & 0x3FF == 789
    tii = wait_for(lambda e: e["nid"] == NID["TII"] and e.get("ids"))[-1]
    assert tii["ids"] == [(21, 5)]                                                    # transmitter pattern 21, comb 5
    assert tii["spec"].shape == (384,) and tii["spec"].max() > 20 * np.median(tii["spec"])
    time.sleep(0.3)
    L.dabsdrRequest_GetEnsemble(handle)
    ens = wait_for(lambda e: e["nid"] == NID["ENSEMBLE_INFO"] and e["status"] == 0)[-1]
    assert ens["ueid"] == 0x00E21234 and ens["lto"] == 2 and ens["label"].startswith("GRAFT ENS") and ens["freq"] == 225648
    L.dabsdrRequest_GetServiceList(handle)
    sl = wait_for(lambda e: e["nid"] == NID["SERVICE_LIST"])[-1]
    assert sorted(s[0] for s in sl["services"]) == [0x1A01, 0x1A02]
    assert all(lbl.startswith("SERVICE 0") for _, lbl in sl["services"])
    # the rest of the start-up sequence of radiocontrol.cpp:1381-1870: components, user applications, announcements
    L.dabsdrRequest_GetServiceComponents.argtypes = [C.c_void_p, C.c_uint32]
    L.dabsdrRequest_GetUserAppList.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8]
    L.dabsdrRequest_GetAnnouncementSupport.argtypes = [C.c_void_p, C.c_uint32]
    L.dabsdrRequest_GetServiceComponents(handle, 0x1A01)
    cl = wait_for(lambda e: e["nid"] == NID["SERVICE_COMPONENT_LIST"] and e.get("sid") == 0x1A01)[-1]
    assert cl["comps"] == [dict(scids=0, subch=0, addr=0, size=48, lang=9, napps=1, tmid=0, ascty=63, kbps=64, ps=2)]
    L.dabsdrRequest_GetUserAppList(handle, 0x1A01, 0)
    ul = wait_for(lambda e: e["nid"] == NID["USER_APP_LIST"] and e.get("sid") == 0x1A01)[-1]
    assert ul["scids"] == 0 and ul["apps"] == [(2, bytes([0x0C, 0x3C]))]                # MOT slide show over X-PAD application type 12
    L.dabsdrRequest_GetAnnouncementSupport(handle, 0x1A01)
    an = wait_for(lambda e: e["nid"] == NID["ANNOUNCEMENT_SUPPORT"] and e.get("sid") == 0x1A01)[-1]
    assert an["asu"] == 0x0002 and an["clusters"] == [7]
    assert wait_for(lambda e: e["nid"] == NID["ANNOUNCEMENT_SWITCHING"])[-1]["asw"] == [(7, 0, 0x0002)]
    assert wait_for(lambda e: e["nid"] == NID["PTY"] and e.get("sid") == 0x1A01)[-1]["s"] == 10
    L.dabsdrRequest_Exit(handle)
    L.dabsdrDeinit.argtypes = [C.POINTER(C.c_void_p)]
    L.dabsdrDeinit(C.byref(handle))
    assert not handle.value


def test_service_selection_delivers_dabplus_access_units():
    """tune -> select the DAB+ service -> the audio callback (dabsdr.h:390) receives exactly the
    transmitted access units (header 0x70 = 48 kHz DAC, SBR, stereo), as radiocontrol.cpp:2542 expects."""
    L = aa.load_library()
    sub = [[0, 0, 3, 64], [48, 1, 4, 32]]
    n_frames = 30
    # every access unit starts with a data_stream_element whose X-PAD carries a dynamic label (three segments, repeated)
    from tests.test_pad import au as pad_au, dl_groups, spread, xpad_var
    dl_text = "GRAFT FM: super frames, Reed-Solomon and PAD on an MI355X"
    heads = []
    for rep in range(8):
        for g in dl_groups(dl_text, toggle=rep & 1):
            heads += [pad_au(xpad_var([sub]), filler=b"") for sub in spread(g, 2, 8)]
    sf_rows, aus_tx = ob.superframes(64, n_frames * 4 // 5, seed=11, au_heads=heads)
    payload = np.zeros((n_frames * 4, 192 + 96), dtype=np.uint8)
    payload[:len(sf_rows), :192] = sf_rows
    payload[:, 192:] = np.random.default_rng(2).integers(0, 256, (n_frames * 4, 96), dtype=np.uint8)
    iq, fib, _ = ob.tx_generate(seed=78, eid=0x1235, n_frames=n_frames, subch=sub, delay=3000, snr_db=22.0, cfo_hz=-800.0, payload=payload)
    samples = iq.astype(np.float32) - 128.0
    pos, gate = [0], threading.Event()
    got, events, lock = [], [], threading.Lock()
    handle = C.c_void_p()

    class AudioCB(C.Structure):
        _fields_ = [("id", C.c_int), ("ASCTy", C.c_uint8), ("header", C.c_uint8), ("auLen", C.c_uint16), ("pAuData", C.POINTER(C.c_uint8))]

    @C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_uint16)
    def get_samples(buf, n):
        if pos[0] >= 2 * 8 * 196608 and not gate.is_set():
            gate.wait(0.05)                                  # pace the un-paced library until the service is selected
        out = np.ctypeslib.as_array(buf, shape=(2 * n,))
        take = samples[pos[0]:pos[0] + 2 * n]
        out[:len(take)] = take
        out[len(take):] = 0.0
        pos[0] += 2 * n

    @C.CFUNCTYPE(None, C.POINTER(Ntf), C.c_void_p)
    def on_ntf(p, ctx):
        with lock:
            events.append((p.contents.nid, p.contents.status))

    class DlCB(C.Structure):
        _fields_ = [("id", C.c_int), ("len", C.c_uint16), ("pData", C.POINTER(C.c_uint8))]

    labels = []

    @C.CFUNCTYPE(None, C.POINTER(DlCB), C.c_void_p)
    def on_dl(p, ctx):
        d = p.contents
        with lock:
            labels.append(bytes(np.ctypeslib.as_array(d.pData, shape=(d.len,))))

    @C.CFUNCTYPE(None, C.POINTER(AudioCB), C.c_void_p)
    def on_audio(p, ctx):
        a = p.contents
        with lock:
            got.append((a.id, a.ASCTy, a.header, bytes(np.ctypeslib.as_array(a.pAuData, shape=(a.auLen,)))))

    L.dabsdrInit.argtypes = [C.POINTER(C.c_void_p)]
    assert L.dabsdrInit(C.byref(handle)) == 0
    for name in ("dabsdrRegisterInputFcn", "dabsdrRegisterDummyInputFcn"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        getattr(L, name)(handle, C.cast(get_samples, C.c_void_p))
    for name, fn in (("dabsdrRegisterNotificationCb", on_ntf), ("dabsdrRegisterAudioCb", on_audio), ("dabsdrRegisterDynamicLabelCb", on_dl)):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        getattr(L, name)(handle, C.cast(fn, C.c_void_p), None)
    L.dabsdr.argtypes = [C.c_void_p]
    L.dabsdrRequest_Tune.argtypes = [C.c_void_p, C.c_uint32]
    L.dabsdrRequest_ServiceSelection.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8, C.c_int]
    L.dabsdrRequest_Exit.argtypes = [C.c_void_p]
    L.dabsdr(handle)
    L.dabsdrRequest_Tune(handle, 225648)

    def wait_until(pred, timeout=60.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            with lock:
                if pred():
                    return
            time.sleep(0.02)
        raise AssertionError(f"timeout: events={events[-8:]} n_au={len(got)}")

    wait_until(lambda: (NID["SYNC_STATUS"], 0) in events)
    time.sleep(0.5)                                          # FIG database fills while the gate holds the input at frame 8
    L.dabsdrRequest_ServiceSelection(handle, 0x1A01, 0, 0)
    wait_until(lambda: (NID["SERVICE_SELECTION"], 0) in events)
    gate.set()
    wait_until(lambda: len(got) >= 24)
    L.dabsdrRequest_Exit(handle)
    L.dabsdrDeinit.argtypes = [C.POINTER(C.c_void_p)]
    L.dabsdrDeinit(C.byref(handle))
    with lock:
        rx = list(got)
    # EVERY unit the callback got: primary decoder, ASCTy 63, header 0x70 — or 0xF0 (conceal bit, dabsdr.h:47-60) for the units of
    # the one super frame the end of the 30-frame recording cuts (the un-paced library runs on into the zeros behind it: its
    # last super frame starts in the signal and ends in silence; the units keep their place with the conceal bit, nothing follows)
    assert all(g[0] == 0 and g[1] == 63 and g[2] in (0x70, 0xF0) for g in rx), [g[:3] for g in rx if g[2] not in (0x70, 0xF0)]
    clean = [g for g in rx if g[2] == 0x70]
    assert len(rx) - len(clean) <= 3 and all(g[2] == 0x70 for g in rx[:len(clean)]), [g[2] for g in rx]
    tx = [a.tobytes() for a in aus_tx]
    first = tx.index(clean[0][3])                            # first AU delivered after the selection
    assert first % 3 == 0                                    # starts on a super frame boundary (3 AUs per super frame)
    assert len(clean) >= 24 and [g[3] for g in clean] == tx[first:first + len(clean)]
    assert first + len(rx) <= len(tx)                        # and nothing beyond what was sent
    # dynamic label segments (dabsdrDynamicLabelCBFunc_t, dabsdr.h:81-86): prefix + characters, assembled as dldecoder.cpp does
    with lock:
        segs = list(labels)
    assert len(segs) >= 4
    start = next(i for i, sg in enumerate(segs) if sg[0] & 0x40)                    # a "first" segment
    msg, i = b"", start
    while True:
        msg += segs[i][2:2 + (segs[i][0] & 0x0F) + 1]
        if segs[i][0] & 0x20:
            break
        i += 1
    assert msg.decode("latin-1") == dl_text


def test_audio_and_packet_data_component_side_by_side():
    """the reference runs an audio decoder and data decoders at once (dabsdrDecoderId_t): select the DAB+ audio component
    and the packet-mode SPI component of the same service; access units and MSC data groups both arrive."""
    from tests.test_packet_mode import frames_of, packets
    L = aa.load_library()
    sub = [[0, 0, 3, 64], [48, 0, 3, 32]]
    n_frames = 30
    sf_rows, aus_tx = ob.superframes(64, n_frames * 4 // 5, seed=21)
    rng = np.random.default_rng(9)
    groups = [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in rng.integers(30, 400, 40)]
    pk = []
    for i, g in enumerate(groups):
        pk += packets(g, 0x155, 48 if i % 2 else 24, ci0=i)
        pk += packets(bytes(rng.integers(0, 256, 50, dtype=np.uint8)), 0x2AA, 24)      # another address on the same sub-channel
    prows = frames_of(pk, 96)[:n_frames * 4]
    payload = np.zeros((n_frames * 4, 192 + 96), dtype=np.uint8)
    payload[:len(sf_rows), :192] = sf_rows
    payload[:len(prows), 192:] = prows
    sent = len(prows)
    iq, _, _ = ob.tx_generate(seed=79, eid=0x1236, n_frames=n_frames, subch=sub, delay=2500, snr_db=22.0, cfo_hz=300.0,
                              payload=payload, packet_sub=1)
    samples = iq.astype(np.float32) - 128.0
    pos, gate = [0], threading.Event()
    aus, dgs, events, lock = [], [], [], threading.Lock()
    handle = C.c_void_p()

    class AudioCB(C.Structure):
        _fields_ = [("id", C.c_int), ("ASCTy", C.c_uint8), ("header", C.c_uint8), ("auLen", C.c_uint16), ("pAuData", C.POINTER(C.c_uint8))]

    class DgCB(C.Structure):
        _fields_ = [("id", C.c_int), ("SCId", C.c_uint16), ("userAppType", C.c_uint16), ("dgLen", C.c_uint16), ("pDgData", C.POINTER(C.c_uint8))]

    @C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_uint16)
    def get_samples(buf, n):
        if pos[0] >= 2 * 8 * 196608 and not gate.is_set():
            gate.wait(0.05)
        out = np.ctypeslib.as_array(buf, shape=(2 * n,))
        take = samples[pos[0]:pos[0] + 2 * n]
        out[:len(take)] = take
        out[len(take):] = 0.0
        pos[0] += 2 * n

    @C.CFUNCTYPE(None, C.POINTER(Ntf), C.c_void_p)
    def on_ntf(p, ctx):
        with lock:
            events.append((p.contents.nid, p.contents.status))

    @C.CFUNCTYPE(None, C.POINTER(AudioCB), C.c_void_p)
    def on_audio(p, ctx):
        a = p.contents
        with lock:
            aus.append((a.id, bytes(np.ctypeslib.as_array(a.pAuData, shape=(a.auLen,)))))

    @C.CFUNCTYPE(None, C.POINTER(DgCB), C.c_void_p)
    def on_dg(p, ctx):
        d = p.contents
        with lock:
            dgs.append((d.id, d.SCId, d.userAppType, bytes(np.ctypeslib.as_array(d.pDgData, shape=(d.dgLen,)))))

    L.dabsdrInit.argtypes = [C.POINTER(C.c_void_p)]
    assert L.dabsdrInit(C.byref(handle)) == 0
    for name in ("dabsdrRegisterInputFcn", "dabsdrRegisterDummyInputFcn"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        getattr(L, name)(handle, C.cast(get_samples, C.c_void_p))
    for name, fn in (("dabsdrRegisterNotificationCb", on_ntf), ("dabsdrRegisterAudioCb", on_audio), ("dabsdrRegisterDataGroupCb", on_dg)):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        getattr(L, name)(handle, C.cast(fn, C.c_void_p), None)
    L.dabsdr.argtypes = [C.c_void_p]
    L.dabsdrRequest_Tune.argtypes = [C.c_void_p, C.c_uint32]
    L.dabsdrRequest_ServiceSelection.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8, C.c_int]
    L.dabsdrRequest_ServiceStop.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8, C.c_int]
    L.dabsdrRequest_Exit.argtypes = [C.c_void_p]
    L.dabsdr(handle)
    L.dabsdrRequest_Tune(handle, 225648)

    def wait_until(pred, timeout=60.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            with lock:
                if pred():
                    return
            time.sleep(0.02)
        raise AssertionError(f"timeout: events={events[-8:]} n_au={len(aus)} n_dg={len(dgs)}")

    wait_until(lambda: (NID["SYNC_STATUS"], 0) in events)
    time.sleep(0.5)
    L.dabsdrRequest_ServiceSelection(handle, 0x1A01, 0, 0)            # audio, primary decoder
    L.dabsdrRequest_ServiceSelection(handle, 0x1A01, 1, -1)           # SCIdS 1: the packet component, data decoder
    wait_until(lambda: events.count((NID["SERVICE_SELECTION"], 0)) == 2)
    gate.set()
    wait_until(lambda: len(aus) >= 18 and len(dgs) >= 8)
    L.dabsdrRequest_ServiceStop(handle, 0x1A01, 1, -1)
    wait_until(lambda: (9, 0) in events)                               # DABSDR_NID_SERVICE_STOP
    L.dabsdrRequest_Exit(handle)
    L.dabsdrDeinit.argtypes = [C.POINTER(C.c_void_p)]
    L.dabsdrDeinit(C.byref(handle))
    with lock:
        got_dg, got_au = list(dgs), list(aus)
    assert all(g[0] == -1 and g[1] == 0x201 and g[2] == 7 for g in got_dg)
    first = groups.index(got_dg[0][3])                                 # first group complete after the selection took effect
    assert [g[3] for g in got_dg] == groups[first:first + len(got_dg)] and sent > 0
    tx = [a.tobytes() for a in aus_tx]
    k0 = tx.index(got_au[0][1])
    n_cmp = min(len(got_au), 45)                                        # before the transmission (30 frames) runs out
    assert n_cmp >= 18 and all(a[0] == 0 for a in got_au) and [a[1] for a in got_au[:n_cmp]] == tx[k0:k0 + n_cmp]   # no gap at the stop
