/*
 * oracle/dab_spec.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the DAB Mode-I physical layer constants and bit-level
 * coding rules from ETSI EN 300 401.  The reference (AbracaDABra) ships this
 * path only as a closed binary (reference: lib/linux_x86_64/libdabsdr.so.4.0.1,
 * header lib/linux_x86_64/dabsdr.h:397-429), so nothing here follows a
 * reference source file: it follows the published standard, using the
 * conventions listed in SURVEY.md Appendix B.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
 * this path and its prebuilt binary is not run in this build
 * (environment rule: prebuilt machine code inside the reference is never
 * loaded).  The oracle is pinned only by the standard's own known answers
 * (tests/test_oracle_spec.py) and by TX->RX round trips.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.
 */
#ifndef DAB_SPEC_H
#define DAB_SPEC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mode I timing, in samples at 2.048 Msps (EN 300 401 §14.2 table 38) */
#define DAB_FS          2048000
#define DAB_TF          196608      /* transmission frame            */
#define DAB_TNULL       2656        /* null symbol                   */
#define DAB_TS          2552        /* OFDM symbol incl. guard       */
#define DAB_TU          2048        /* useful part = FFT size        */
#define DAB_TG          504         /* guard interval                */
#define DAB_NSYM        76          /* symbols per frame, excl. null */
#define DAB_K           1536        /* active carriers               */
#define DAB_SYM_BITS    3072        /* 2 bits per carrier            */
#define DAB_FIC_SYMS    3
#define DAB_FIC_BITS    9216
#define DAB_FIC_CW      4           /* punctured codewords per frame */
#define DAB_FIC_CW_BITS 2304
#define DAB_FIC_CW_IN   768         /* 3 FIBs of 256 bits            */
#define DAB_FIB_BYTES   32
#define DAB_FIBS_PER_FRAME 12
#define DAB_CIF_BITS    55296       /* 864 CU x 64 bit               */
#define DAB_CIF_SYMS    18
#define DAB_CIFS        4           /* CIFs per frame                */
#define DAB_CU_BITS     64
#define DAB_NCU         864
#define DAB_TI_DEPTH    16          /* time interleaver span in CIFs */

/* §14.6: carrier index k_n (−768..768, ≠0) of QPSK symbol n, n = 0..1535 */
void dab_freq_interleaver(int16_t k_of_n[DAB_K]);

/* §14.3.2: phase reference symbol. q[bin] ∈ {0,1,2,3} = phase in units of
 * pi/2 on FFT bin (k mod 2048); −1 on unused bins. */
void dab_prs_quadrants(int8_t q[DAB_TU]);

/* §11.1.2 table 29: puncturing vector for index PI = 1..24 as 32 flags */
void dab_punct_vector(int pi, uint8_t v[32]);

/* §12: time interleaver delay (in CIFs) for bit index i: d(i mod 16) */
int dab_ti_delay(int i);

/* §10: energy dispersal PRBS x^9+x^5+1, all-ones start; n output bits */
void dab_prbs(uint8_t *bits, int n);

/* §5.2.1: FIB / AU CRC-16 (poly 0x1021, init 0xFFFF, result inverted) */
uint16_t dab_crc16(const uint8_t *data, int nbytes);

/* §11.1.1: mother code, generators 133,171,145,133 (octal).  in: nin bits
 * (one per byte); out: 4*(nin+6) bits (tail of 6 zero bits appended). */
void dab_conv_encode(const uint8_t *in, int nin, uint8_t *out);

/* encoder output nibble (x0 in bit 3 … x3 in bit 0) for a state
 * (bit 5 = newest input bit … bit 0 = oldest) and an input bit */
int dab_conv_output(int state, int in);

/* Protection profile: up to 4 segments of L blocks (128 mother bits each)
 * punctured with index PI, followed by the 24-bit tail punctured to 12. */
typedef struct {
    int nseg;
    int L[4];
    int PI[4];
    int n_in;      /* information bits per codeword (without tail)   */
    int n_coded;   /* punctured bits per codeword                     */
    int n_cu;      /* sub-channel size in capacity units (0 for FIC) */
} dab_profile_t;

void dab_profile_fic(dab_profile_t *p);
/* §11.3.2 EEP: option 0 = set A (bitrate multiple of 8 kbit/s), 1 = set B
 * (multiple of 32); level 1..4.  Returns 0 on success. */
int dab_profile_eep(int option, int level, int bitrate_kbps, dab_profile_t *p);

/* §11.3.1 UEP: index = 6-bit table index of FIG 0/1 short form (0..63) */
int dab_profile_uep(int index, dab_profile_t *p, int *kbps, int *level);
/* option 0/1 = EEP set A/B (level 1..4, kbps); option 2 = UEP with level = table index */
int dab_profile_any(int option, int level, int kbps, dab_profile_t *p);

/* puncture / depuncture map: for every trellis step t (0 … n_in+5) the
 * offset of its first kept bit in the punctured stream and the 4-bit keep
 * mask (bit 3 = x0).  info[t] = (offset << 4) | mask.  Returns step count. */
int dab_profile_stepinfo(const dab_profile_t *p, uint32_t *info);

/* puncture 4*(n_in+6) mother bits → n_coded bits */
int dab_puncture(const dab_profile_t *p, const uint8_t *mother, uint8_t *coded);

#ifdef __cplusplus
}
#endif
#endif
