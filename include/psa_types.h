/*
 * psa_types.h — data contract of the offset x mutation alignment search.
 *
 * These are the two structs and the constants that cross the host/device
 * boundary of the reference program.  Field order and types are kept so a
 * caller built against the reference headers can pass its objects straight
 * through:
 *
 *   ProgramData   <- reference program_data.h:6-11
 *   Mutant        <- reference mutant.h:6-10
 *   sign chars    <- reference def.h:4-8, NOT_FOUND_CHAR def.h:10
 *   weight order  <- reference def.h:12-15
 *   file names    <- reference def.h:20-21
 *   limits        <- reference def.h:35-36 (10000+1 / 5000+1, unparenthesised
 *                    there).  Here they are parenthesised and overridable:
 *                    the default is the BASELINE.json size (1M / 10k); build
 *                    with -DPSA_SEQ1_MAX_LEN=10001 -DPSA_SEQ2_MAX_LEN=5001 to
 *                    get the reference's exact 15,048-byte ProgramData.
 *                    The library SYMBOL gpu_run_program always reads the
 *                    reference's stock layout (PSA_REF_SEQ*_MAX_LEN); psa.h
 *                    routes callers built with other limits through
 *                    gpu_run_program_layout.
 */
#ifndef PSA_TYPES_H
#define PSA_TYPES_H

#ifdef __cplusplus
extern "C" {
#endif

/* the reference's stock limits (def.h:35-36): sizeof(ProgramData) == 15,048 with these */
#define PSA_REF_SEQ1_MAX_LEN (10000 + 1)
#define PSA_REF_SEQ2_MAX_LEN (5000 + 1)

#ifndef PSA_SEQ1_MAX_LEN
#define PSA_SEQ1_MAX_LEN (1000000 + 1)
#endif
#ifndef PSA_SEQ2_MAX_LEN
#define PSA_SEQ2_MAX_LEN (10000 + 1)
#endif

/* the longest Seq1 the library takes (all lengths and offsets of the C ABI are int; inside, padded lengths, tile counts
 * and window ends are 32-bit too — 2^30 leaves them a factor of two).  Longer: PSA_ERR_ARG. */
#define PSA_MAX_SEQ1 (1 << 30)

/* pair signs and the weight each one selects */
#define PSA_SIGN_STAR   '*'   /* identical residues            -> +w[0] */
#define PSA_SIGN_COLON  ':'   /* same conservative group       -> -w[1] */
#define PSA_SIGN_DOT    '.'   /* same semi-conservative group  -> -w[2] */
#define PSA_SIGN_SPACE  '_'   /* unrelated                     -> -w[3] */
#define PSA_HYPHEN      '-'
#define PSA_NO_CHAR     '\0'

#define PSA_NUM_WEIGHTS 4
#define PSA_ALPHABET    26    /* 'A'..'Z'; '-' is the 27th symbol */

#define PSA_INPUT_FILE  "./input.txt"
#define PSA_OUTPUT_FILE "./output.txt"
#define PSA_MAXIMUM_STR "maximum"
#define PSA_MINIMUM_STR "minimum"

/* Input bundle (reference program_data.h:6-11). NUL-terminated sequences. */
typedef struct _data {
    int    is_max;
    double weights[PSA_NUM_WEIGHTS];
    char   seq1[PSA_SEQ1_MAX_LEN];
    char   seq2[PSA_SEQ2_MAX_LEN];
} ProgramData;

/* Result (reference mutant.h:6-10): offset is absolute into seq1,
 * char_offset indexes seq2, ch is the substituted character. */
typedef struct _mutant {
    int  offset;
    int  char_offset;
    char ch;
} Mutant;

#ifdef __cplusplus
}
#endif
#endif /* PSA_TYPES_H */
