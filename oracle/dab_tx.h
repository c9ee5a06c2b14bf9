/*
 * oracle/dab_tx.h — TEST INFRASTRUCTURE ONLY: configuration of the synthetic transmitter (dab_tx.c).
 */
#ifndef DAB_TX_H
#define DAB_TX_H
#include <stdint.h>

typedef struct {
    uint64_t seed;
    int32_t  eid;          /* ensemble id                              */
    int32_t  n_frames;
    int32_t  n_subch;      /* sub-channels; cfg in subch[] below       */
    int32_t  delay;        /* noise-only samples before frame 0        */
    int32_t  loop;         /* 1: periodic signal, TI history wraps     */
    int32_t  fmt;          /* 0: u8 IQ, 1: s16 IQ                      */
    double   snr_db;       /* >= 100: noiseless                        */
    double   cfo_hz;
    double   rms;          /* complex RMS of the signal in LSB         */
    int32_t  subch[64][4]; /* {start_cu, option(0=A,1=B,2=UEP), level, kbps} */
    int32_t  payload_given;/* 1: msc_out already holds the payload to transmit    */
    int32_t  tii_main, tii_sub; /* TII in every null symbol (EN 300 401 §14.8); main < 0: none */
    int32_t  extra_figs;   /* 1: also send FIG 0/5, 0/8, 0/13, 0/17, 0/18, 0/19 for the first service */
    int32_t  packet_sub;   /* >= 1: this sub-channel is a packet-mode data component (SCId 0x200 + index, packet address 0x155,
                              user application 7 = SPI) of the FIRST service, announced by FIG 0/2 (TMId 3), 0/3 and 0/13; 0 = none */
    /* channel impairments (all 0 = ideal channel) */
    double   sco_ppm;      /* sampling clock offset of the receiver's ADC: the recording holds the signal sampled at
                              Fs * (1 + sco_ppm 1e-6) (windowed-sinc interpolation); not with loop = 1 */
    double   dc_i, dc_q;   /* DC offset in LSB added before quantisation                                */
    double   echo_db;      /* second path: attenuation in dB (> 0), 0 = no echo                          */
    double   echo_phase;   /* its carrier phase in radians                                              */
    int32_t  echo_delay;   /* its delay in samples (inside the 504-sample guard interval for a benign channel) */
    int32_t  eid2_from;    /* >= 1: frames from this index on belong to another ensemble (EId + 1): two ensembles
                              concatenated in one recording (RESET(NEW_EID) test) */
} dab_tx_cfg_t;

int dab_tx_msc_bytes_per_cif(const dab_tx_cfg_t *c);
int dab_tx_generate(const dab_tx_cfg_t *c, void *iq, uint8_t *fib_out, uint8_t *msc_out);
#endif
