/* Host-side scene-graph loader: GQA scene-graph JSON -> token tensors -> collated PyG-style batch.
 *
 * C ABI of libisg_loader.so (plain C++17, no GPU dependency): what a binding of the reference's data path would call in
 * place of
 *   GQASceneGraphs.build_scene_graph_encoding_vocab   ISubGVQA/datasets/scene_graph.py:145-183   isg_sg_vocab_build
 *   json.load of the scene-graph files                 ISubGVQA/datasets/scene_graph.py:55-66     isg_sg_store_add_json*
 *   GQASceneGraphs.convert_one_gqa_scene_graph         ISubGVQA/datasets/scene_graph.py:199-389   (at add time, once per image)
 *   GQASceneGraphs.query_and_translate                 ISubGVQA/datasets/scene_graph.py:71-143    isg_sg_collate* (per id)
 *   GQADataset.__getitem__ squeeze + gqa_collate's
 *   Batch.from_data_list                               ISubGVQA/datasets/gqa.py:170-175,258       isg_sg_collate
 * The caller owns every output buffer (pin it if it is to be copied to the GPU asynchronously); all tensors are int64
 * like the reference's.  Every function returns 0 on success, a negative ISG_LD_* code otherwise;
 * isg_sg_last_error() describes the last failure of the calling thread.
 */
#ifndef ISG_LOADER_H
#define ISG_LOADER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISG_LOADER_ABI_VERSION 1

#define ISG_LD_OK 0
#define ISG_LD_EINVAL (-1)  /* NULL pointer / negative size */
#define ISG_LD_EPARSE (-2)  /* malformed JSON or a scene graph that references an unknown object id */
#define ISG_LD_EIO (-3)     /* file cannot be read */

typedef struct isg_sg_vocab isg_sg_vocab;
typedef struct isg_sg_store isg_sg_store;

int isg_loader_abi_version(void);
const char *isg_sg_last_error(void);

/* tokens: the concatenation name_gqa + attr_gqa + rel_gqa + objects + predicates + attributes, in that order
 * (scene_graph.py:152-163); "<self>" and "pokemon" are appended here (:164-165).  Applies torchtext.vocab.vocab's rules
 * to the reference's {token: position} dict: specials <unk> <pad> <sos> <eos> <self> first, duplicates keep their first
 * place, and the token whose last position is 0 is dropped (its 'frequency' is 0 < min_freq). */
int isg_sg_vocab_build(const char *const *tokens, int64_t n_tokens, isg_sg_vocab **out);
int64_t isg_sg_vocab_size(const isg_sg_vocab *v);
int64_t isg_sg_vocab_lookup(const isg_sg_vocab *v, const char *token); /* index, or -1 when absent */
void isg_sg_vocab_free(isg_sg_vocab *v);

/* A store holds the converted graphs of any number of JSON files; an image id added twice keeps the LAST version (the
 * reference merges its three files with dict `|`, scene_graph.py:68-72).  The vocab must outlive the store. */
int isg_sg_store_create(const isg_sg_vocab *v, isg_sg_store **out);
int isg_sg_store_add_json(isg_sg_store *s, const char *text, int64_t len);
int isg_sg_store_add_json_file(isg_sg_store *s, const char *path);
int64_t isg_sg_store_num_graphs(const isg_sg_store *s);
int64_t isg_sg_store_find(const isg_sg_store *s, const char *image_id); /* slot, or -1 (-> the 6-node dummy graph) */
void isg_sg_store_free(isg_sg_store *s);

/* Slots of B image ids (-1 = not in the store), so a dataset resolves its ids once, not per batch. */
int isg_sg_store_find_many(const isg_sg_store *s, const char *const *image_ids, int64_t B, int64_t *slots);

/* Batch of B slots (slot -1 and graphs with a single edge become the reference's 6-node dummy graph).
 * sizes: totals[0..2] = nodes, edges, added symmetric edges.
 * collate: x[N,4], edge_index[2,E] (row 0 sources, row 1 targets, node ids offset per graph), edge_attr[E], x_bbox[N,4],
 * added_sym_edge[S] (per-graph edge positions, NOT offset: quirk Q6), batch[N], ptr[B+1];
 * bounds[0] = max nodes per graph, bounds[1] = max edges per graph (GraphPlan hints: no device sync needed).
 * n_threads > 1 copies disjoint graph ranges (balanced by edge count) on that many threads. */
int isg_sg_collate_sizes(const isg_sg_store *s, const int64_t *slots, int64_t B, int64_t *totals);
int isg_sg_collate(const isg_sg_store *s, const int64_t *slots, int64_t B, int64_t *x, int64_t *edge_index,
                   int64_t *edge_attr, int64_t *x_bbox, int64_t *added_sym_edge, int64_t *batch, int64_t *ptr,
                   int64_t *bounds, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
