"""A minimal host for the reference's 24-function dabsdr API (test helper): what src/radiocontrol.cpp does with
the library — init, register callbacks, dabsdr(), requests, notifications — over a raw-file style input."""
import ctypes as C
import threading
import time

import numpy as np

import abracadabra_amd as aa
from test_gpu_legacy_api import NID, CompItem, CompList, Ensemble, Ntf, Periodic, ServiceItem, ServiceList

TF = 196608


class LegacyHost:
    def __init__(self, samples_f32, gate_at=None, gate_on_new_eid=False, pace=0.0):
        """samples_f32: interleaved I,Q floats the input callback hands out (zeros after the end, like a flushed FIFO).
        gate_at: sample count (complex) after which the input blocks until open_gate() (the library is un-paced).
        pace: seconds every input call takes at least — 0.0025 (30 ms per frame) makes the library decode frame by frame and
        deliver at once, as it does behind the reference's raw-file input, which a 50 ms timer paces (rawfileinput.cpp:236-238)."""
        self.pace = pace
        self.L = aa.load_library()
        self.samples = np.ascontiguousarray(samples_f32, dtype=np.float32)
        self.pos = 0
        self.gate_at = gate_at
        self.gate_on_new_eid = gate_on_new_eid      # hold the input from RESET(NEW_EID) on until open_gate(): the host's restart takes time
        self.gate = threading.Event()
        self.events, self.lock = [], threading.Lock()
        self.handle = C.c_void_p()
        L = self.L

        @C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_uint16)
        def get_samples(buf, n):
            if self.gate_at is not None and self.pos >= 2 * self.gate_at and not self.gate.is_set():
                self.gate.wait(0.05)
            if self.pace:
                time.sleep(self.pace)
            out = np.ctypeslib.as_array(buf, shape=(2 * n,))
            take = self.samples[self.pos:self.pos + 2 * n]
            out[:len(take)] = take
            out[len(take):] = 0.0
            self.pos += 2 * n

        @C.CFUNCTYPE(None, C.POINTER(Ntf), C.c_void_p)
        def on_ntf(p, ctx):
            n = p.contents
            rec = dict(nid=n.nid, status=n.status, len=n.len, at=self.pos // 2)
            if n.nid == NID["TUNE"]:
                rec["freq"] = C.cast(n.pData, C.POINTER(C.c_uint32)).contents.value
            elif n.nid == NID["RESET"]:
                rec["flag"] = C.cast(n.pData, C.POINTER(C.c_int)).contents.value
                if rec["flag"] == 1 and self.gate_on_new_eid:
                    self.gate_at = self.pos // 2
            elif n.nid == NID["SYNC_STATUS"]:
                rec["level"] = C.cast(n.pData, C.POINTER(C.c_int)).contents.value
                rec["snr10"] = C.cast(n.pData + 4, C.POINTER(C.c_int16)).contents.value
            elif n.nid == NID["ENSEMBLE_INFO"]:
                e = C.cast(n.pData, C.POINTER(Ensemble)).contents
                rec.update(ueid=e.ueid, lto=e.LTO, label=e.label.str.decode(), charField=e.label.charField, freq=e.frequency)
            elif n.nid == NID["PERIODIC"] and n.pData:
                pr = C.cast(n.pData, C.POINTER(Periodic)).contents
                rec.update(fib_err=pr.fibErrorCntr, foff=pr.freqOffset, level=pr.syncLevel, snr10=pr.snr10, crc_ok=pr.mscCrcOkCntr,
                           crc_err=pr.mscCrcErrorCntr, rs_unc=pr.rsUncorrectableCntr, audio_bytes=pr.audioServiceBytes)
            elif n.nid == NID["SERVICE_LIST"]:
                sl = C.cast(n.pData, C.POINTER(ServiceList)).contents
                items = []
                for i in range(sl.numServices):
                    it = ServiceItem()
                    sl.getItem(self.handle, i, C.byref(it))
                    items.append(dict(sid=it.sid, label=it.label.str.decode(), pty=(it.pty_s, it.pty_d)))
                rec["services"] = items
            elif n.nid == NID["SERVICE_COMPONENT_LIST"]:
                cl = C.cast(n.pData, C.POINTER(CompList)).contents
                comps = []
                for i in range(cl.numServiceComponents):
                    it = CompItem()
                    cl.getItem(self.handle, i, C.byref(it))
                    comps.append(dict(scids=it.SCIdS, subch=it.SubChId, addr=it.SubChAddr, size=it.SubChSize, prot=it.protectionLevel, ps=it.ps,
                                      tmid=it.TMId, ascty=it.u.streamAudio.ASCTy, kbps=it.u.streamAudio.bitRate, fec=it.uepIdx))   # uepIdx / fecScheme: one union
                rec.update(sid=cl.SId, comps=comps)
            with self.lock:
                self.events.append(rec)

        class DgCB(C.Structure):                      # dabsdrDataGroupCBData_t (dabsdr.h:89-96)
            _fields_ = [("id", C.c_int), ("SCId", C.c_uint16), ("userAppType", C.c_uint16), ("dgLen", C.c_uint16), ("pDgData", C.POINTER(C.c_uint8))]

        self.data_groups = []

        @C.CFUNCTYPE(None, C.POINTER(DgCB), C.c_void_p)
        def on_dg(p, ctx):
            d = p.contents
            with self.lock:
                self.data_groups.append((d.id, d.SCId, d.userAppType, bytes(np.ctypeslib.as_array(d.pDgData, shape=(d.dgLen,)))))

        class AudioCB(C.Structure):                   # dabsdrAudioCBData_t (dabsdr.h:71-78)
            _fields_ = [("id", C.c_int), ("ASCTy", C.c_uint8), ("header", C.c_uint8), ("auLen", C.c_uint16), ("pAuData", C.POINTER(C.c_uint8))]

        class DlCB(C.Structure):                      # dabsdrDynamicLabelCBData_t (dabsdr.h:81-86)
            _fields_ = [("id", C.c_int), ("len", C.c_uint16), ("pData", C.POINTER(C.c_uint8))]

        self.audio, self.labels, self.audio_at = [], [], []

        @C.CFUNCTYPE(None, C.POINTER(AudioCB), C.c_void_p)
        def on_audio(p, ctx):
            a = p.contents
            with self.lock:
                self.audio.append((a.id, a.ASCTy, a.header, bytes(np.ctypeslib.as_array(a.pAuData, shape=(a.auLen,)))))
                self.audio_at.append(self.pos // 2)

        @C.CFUNCTYPE(None, C.POINTER(DlCB), C.c_void_p)
        def on_dl(p, ctx):
            d = p.contents
            with self.lock:
                self.labels.append(bytes(np.ctypeslib.as_array(d.pData, shape=(d.len,))))

        self.spectra = []

        @C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_void_p)
        def on_spectrum(p, kind, ctx):                # dabsdrSpectrumCBFunc_t: 2048 floats, borrowed (dabsdr.h:393)
            with self.lock:
                self.spectra.append((kind, np.ctypeslib.as_array(p, shape=(2048,)).copy()))

        self._keep = (get_samples, on_ntf, on_dg, on_audio, on_dl, on_spectrum)
        L.dabsdrInit.argtypes = [C.POINTER(C.c_void_p)]
        assert L.dabsdrInit(C.byref(self.handle)) == 0
        for name in ("dabsdrRegisterInputFcn", "dabsdrRegisterDummyInputFcn"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, name)(self.handle, C.cast(get_samples, C.c_void_p))
        L.dabsdrRegisterNotificationCb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabsdrRegisterNotificationCb(self.handle, C.cast(on_ntf, C.c_void_p), None)
        L.dabsdrRegisterDataGroupCb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabsdrRegisterDataGroupCb(self.handle, C.cast(on_dg, C.c_void_p), None)
        for name, fn in (("dabsdrRegisterAudioCb", on_audio), ("dabsdrRegisterDynamicLabelCb", on_dl), ("dabsdrRegisterSignalSpectrumCb", on_spectrum)):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(L, name)(self.handle, C.cast(fn, C.c_void_p), None)
        for name in ("dabsdr", "dabsdrRequest_GetEnsemble", "dabsdrRequest_GetServiceList", "dabsdrRequest_Exit"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.dabsdrRequest_Tune.argtypes = [C.c_void_p, C.c_uint32]
        L.dabsdrRequest_SetPeriodicNotify.argtypes = [C.c_void_p, C.c_uint8, C.c_uint32]
        L.dabsdrRequest_GetServiceComponents.argtypes = [C.c_void_p, C.c_uint32]
        L.dabsdrRequest_ServiceSelection.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8, C.c_int]
        L.dabsdr(self.handle)

    def tune(self, khz=225648, periodic=0, period_log2=1):
        """period_log2: periodic notification every 2^n frames (the reference's host asks for 3: radiocontrol.h:43-46)"""
        self.L.dabsdrRequest_SetPeriodicNotify(self.handle, period_log2, periodic)
        self.L.dabsdrRequest_Tune(self.handle, khz)

    def open_gate(self):
        self.gate.set()

    def wait_for(self, pred, timeout=60.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            with self.lock:
                hit = [e for e in self.events if pred(e)]
            if hit:
                return hit
            time.sleep(0.02)
        with self.lock:
            raise AssertionError(f"timeout; last events: {self.events[-10:]}")

    def close(self):
        if self.handle:
            self.L.dabsdrRequest_Exit(self.handle)
            self.L.dabsdrDeinit.argtypes = [C.POINTER(C.c_void_p)]
            self.L.dabsdrDeinit(C.byref(self.handle))
            assert not self.handle.value
            self.handle = None
