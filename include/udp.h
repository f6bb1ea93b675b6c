/*
 * udp.h -- UDP transport helpers exported by libcassiemujoco.so because the ctypes wrapper binds them
 * (reference include/udp.h:20-60, src/udp.c:58-265).  Outside the physics hot path.
 */
#ifndef UDP_H
#define UDP_H

#include <sys/socket.h>
#include <sys/types.h>

#define PACKET_HEADER_LEN 2

/* sequence / delay bookkeeping of the 2-byte packet header */
typedef struct {
    char seq_num_out;
    char seq_num_in_last;
    char delay;
    char seq_num_in_diff;
} packet_header_info_t;

#ifdef __cplusplus
extern "C" {
#endif
void process_packet_header(packet_header_info_t *info, const unsigned char *header_in, unsigned char *header_out);
int udp_init_host(const char *addr_str, const char *port_str);
int udp_init_client(const char *remote_addr_str, const char *remote_port_str, const char *local_addr_str,
                    const char *local_port_str);
void udp_close(int sock);
ssize_t get_newest_packet(int sock, void *recvbuf, size_t recvlen, struct sockaddr *src_addr, socklen_t *addrlen);
ssize_t wait_for_packet(int sock, void *recvbuf, size_t recvlen, struct sockaddr *src_addr, socklen_t *addrlen);
ssize_t send_packet(int sock, void *sendbuf, size_t sendlen, struct sockaddr *dst_addr, socklen_t addrlen);
#ifdef __cplusplus
}
#endif
#endif
