/* fastq_reader.c -- see fastq_reader.h.  Block reader over zlib (transparent for plain files) with line-level
 * memchr scanning; sequences of multi-line records are concatenated; FASTA and FASTQ records may be mixed. */
#include "fastq_reader.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define TJR_BLOCK (4u << 20)

typedef struct { char *s; size_t len, cap; } tjr_text;

struct tjr_reader
{
  gzFile f;
  unsigned char *blk;
  size_t pos, end;
  int drained;          /* the last gzread came back short: nothing more after this block */
  int marker_seen;      /* the '>' / '@' of the next record has already been consumed */
  int in_memory;        /* blk is the caller's (whole input, nothing to refill or free) */
  size_t marker_pos;    /* offset in blk of the marker of the next record (marker_seen) / of the record returned last */
  size_t rec_start;
  int rec_open;         /* the last tjr_next() found a record marker (rec_start is that record's) */
  tjr_text seq, qual;
};

static int
tjr_refill (tjr_reader *r)
{
  int got;
  if (r->drained) return 0;
  got = gzread (r->f, r->blk, TJR_BLOCK);
  r->pos = 0;
  r->end = got > 0 ? (size_t) got : 0;
  if (got < (int) TJR_BLOCK) r->drained = 1;
  return r->end > 0;
}

static inline int
tjr_byte (tjr_reader *r)
{
  if (r->pos >= r->end && !tjr_refill (r)) return -1;
  return r->blk[r->pos++];
}

static void
tjr_push (tjr_text *t, const unsigned char *p, size_t n)
{
  if (t->len + n + 1 > t->cap) {
    t->cap = (t->len + n + 1) * 2 + 256;
    t->s = (char *) realloc (t->s, t->cap);
  }
  memcpy (t->s + t->len, p, n);
  t->len += n;
}

/* consume up to and including the next '\n'.  t != NULL: append the line (without '\n'), then drop one trailing
 * '\r' if the accumulated text is longer than one byte.  Returns -1 if the input was already exhausted, else 0. */
static int
tjr_line (tjr_reader *r, tjr_text *t)
{
  if (r->pos >= r->end && r->drained) return -1;
  for (;;) {
    unsigned char *from, *nl;
    if (r->pos >= r->end && !tjr_refill (r)) break;
    from = r->blk + r->pos;
    nl = (unsigned char *) memchr (from, '\n', r->end - r->pos);
    if (nl) {
      if (t) tjr_push (t, from, (size_t) (nl - from));
      r->pos = (size_t) (nl - r->blk) + 1;
      break;
    }
    if (t) tjr_push (t, from, r->end - r->pos);
    r->pos = r->end;
  }
  if (t && t->len > 1 && t->s[t->len - 1] == '\r') t->len--;
  return 0;
}

tjr_reader *
tjr_open (const char *path)
{
  tjr_reader *r;
  gzFile f = gzopen (path, "r");
  if (!f) return NULL;
  gzbuffer (f, 1u << 20);
  r = (tjr_reader *) calloc (1, sizeof (tjr_reader));
  r->f = f;
  r->blk = (unsigned char *) malloc (TJR_BLOCK);
  return r;
}

tjr_reader *
tjr_open_mem (const unsigned char *data, size_t n_bytes, size_t start)
{
  tjr_reader *r = (tjr_reader *) calloc (1, sizeof (tjr_reader));
  r->blk = (unsigned char *) data;
  r->pos = start < n_bytes ? start : n_bytes;
  r->end = n_bytes;
  r->drained = 1;
  r->in_memory = 1;
  return r;
}

size_t tjr_record_start (const tjr_reader *r) { return r->rec_start; }
int tjr_record_open (const tjr_reader *r) { return r->rec_open; }
int tjr_at_end (const tjr_reader *r) { return r->drained && r->pos >= r->end; }

void
tjr_close (tjr_reader *r)
{
  if (!r) return;
  if (!r->in_memory) { gzclose (r->f); free (r->blk); }
  free (r->seq.s); free (r->qual.s); free (r);
}

long
tjr_next (tjr_reader *r, const char **seq)
{
  int c;
  unsigned char ch;
  r->rec_open = 0;
  if (!r->marker_seen) {                       /* hunt for the next record marker, wherever it is */
    do c = tjr_byte (r); while (c != -1 && c != '>' && c != '@');
    if (c == -1) return -1;
    r->marker_pos = r->pos - 1;
  }
  r->rec_start = r->marker_pos;
  r->rec_open = 1;
  r->marker_seen = 0;
  r->seq.len = r->qual.len = 0;
  if (tjr_line (r, NULL) < 0) return -1;       /* header line: name and comment are not needed */
  if (r->in_memory) {
    /* The usual FASTQ record -- one sequence line, a '+' line, one quality line, all of it inside the block -- without
     * copying a byte: the sequence is handed out where it lies (the block is the caller's and outlives the call) and the
     * quality line is only measured.  Everything is looked at before anything is changed; whatever does not fit the
     * pattern (FASTA, wrapped lines, a record cut by the block's end, a blank line) takes the general path below, which
     * this one follows step by step: a line loses one trailing '\r' if it is longer than one byte, and the record is
     * good if the quality line is exactly as long as the sequence. */
    const unsigned char *b = r->blk, *e = r->blk + r->end, *s0 = b + r->pos, *n1, *n2, *n3;
    if (s0 < e && *s0 != '>' && *s0 != '+' && *s0 != '@' && *s0 != '\n' &&
        (n1 = (const unsigned char *) memchr (s0, '\n', (size_t) (e - s0))) != NULL && n1 + 1 < e && n1[1] == '+' &&
        (n2 = (const unsigned char *) memchr (n1 + 1, '\n', (size_t) (e - n1 - 1))) != NULL && n2 + 1 < e &&
        (n3 = (const unsigned char *) memchr (n2 + 1, '\n', (size_t) (e - n2 - 1))) != NULL) {
      size_t ls = (size_t) (n1 - s0), lq = (size_t) (n3 - n2 - 1);
      if (ls > 1 && n1[-1] == '\r') ls--;
      if (lq > 1 && n3[-1] == '\r') lq--;
      if (lq >= ls) {                          /* (a shorter quality line: the general path reads on) */
        r->pos = (size_t) (n3 - b) + 1;
        *seq = (const char *) s0;
        return (lq == ls) ? (long) ls : -2;
      }
    }
  }
  for (;;) {                                   /* sequence lines until a line starts with '+', '>' or '@' */
    c = tjr_byte (r);
    if (c == -1 || c == '>' || c == '+' || c == '@') break;
    if (c == '\n') continue;
    ch = (unsigned char) c;
    tjr_push (&r->seq, &ch, 1);
    tjr_line (r, &r->seq);
  }
  if (c == '>' || c == '@') { r->marker_seen = 1; r->marker_pos = r->pos - 1; }
  if (!r->seq.s) tjr_push (&r->seq, (const unsigned char *) "", 0);
  *seq = r->seq.s;
  if (c != '+') return (long) r->seq.len;      /* FASTA record (or end of input) */
  do c = tjr_byte (r); while (c != -1 && c != '\n');   /* rest of the '+' line */
  if (c == -1) return -2;
  while (tjr_line (r, &r->qual) >= 0 && r->qual.len < r->seq.len) ;
  return (r->qual.len == r->seq.len) ? (long) r->seq.len : -2;
}
